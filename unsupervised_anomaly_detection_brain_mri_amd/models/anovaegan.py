"""models/anovaegan.py:10-80 — AnoVAE-GAN: VAE encoder (mu / log-sigma heads) -> LayerNorm-HW generator with a linear output ->
LayerNorm-HW critic that compares the reconstruction with the input.  The graph itself lives in csrc/uad_gan.hip (uad_gan_create
with UAD_GAN_ANOVAEGAN)."""


def anovaegan(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('anovaegan() is a network descriptor for the HIP engine; pass it as network= to trainers.AnoVAEGAN')


anovaegan.arch = 'AnoVAEGAN'
anovaegan.variant = 'anovaegan'
anovaegan.output_keys = ('z_mu', 'z_log_sigma', 'z_sigma', 'out', 'd_fake_features', 'd_', 'd_features', 'd', 'x_hat',
                         'd_hat_features', 'd_hat')        # anovaegan.py:31-33,50,60-61,67-68,73,78-79
