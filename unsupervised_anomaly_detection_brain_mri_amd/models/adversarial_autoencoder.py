"""models/adversarial_autoencoder.py:10-72 -- network descriptor; the graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE), trained by
trainers/AAE.py."""


def adversarial_autoencoder(*args, **kw):
    raise RuntimeError('adversarial_autoencoder() is a network descriptor for the HIP engine; pass it as network= to trainers.AAE')


adversarial_autoencoder.arch = 'AAE'
adversarial_autoencoder.output_keys = ('z_', 'x_hat', 'd_', 'd', 'z_hat', 'd_hat')
