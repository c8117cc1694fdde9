"""models/fanogan.py:11-84 — unified f-AnoGAN: Encoder (conv blocks + 1x1 conv + Dense -> tanh), Generator (Dense + 1x1 conv +
LayerNorm-HW ConvT blocks -> sigmoid), Discriminator (LayerNorm-HW conv blocks + per-location Dense(1)).  The graph itself
lives in csrc/uad_gan.hip (uad_gan_create)."""


def fanogan(z=None, x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('fanogan() is a network descriptor for the HIP engine; pass it as network= to trainers.fAnoGAN')


fanogan.arch = 'fAnoGAN'
fanogan.variant = 'unified'
# fanogan.py:30,42,47,57-58,64-65,69,75-76,82-83
fanogan.output_keys = ('z_enc', 'x_enc', 'x_', 'd_fake_features', 'd_', 'd_features', 'd', 'x_hat', 'd_hat_features', 'd_hat',
                       'd_enc_features', 'd_enc')
