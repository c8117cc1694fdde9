"""models/fanogan_schlegl.py:11-161 — f-AnoGAN with the ResNet generator / critic of Schlegl et al.: pre-activation residual
blocks (LayerNorm-HW -> ReLU -> k3 conv -> LayerNorm-HW -> ReLU -> k3 conv / ConvT), avg-pool / k1 s2 shortcuts, tanh output; the
encoder is the unified conv encoder + Dense -> tanh.  The graph itself lives in csrc/uad_gan.hip (uad_gan_create with
UAD_GAN_RESNET)."""


def fanogan_schlegl(z=None, x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('fanogan_schlegl() is a network descriptor for the HIP engine; pass it as network= to trainers.fAnoGAN')


fanogan_schlegl.arch = 'fAnoGAN'
fanogan_schlegl.variant = 'resnet'
fanogan_schlegl.output_keys = ('z_enc', 'x_', 'x_enc', 'd_fake_features', 'd_', 'd_features', 'd', 'x_hat', 'd_hat_features', 'd_hat',
                               'd_enc_features', 'd_enc')        # fanogan_schlegl.py:22,58,61,101,104,109,111,114
