"""models/constrained_adversarial_autoencoder.py:10-79 -- network descriptor; the graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE), trained by
trainers/ConstrainedAAE.py."""


def constrained_adversarial_autoencoder(*args, **kw):
    raise RuntimeError('constrained_adversarial_autoencoder() is a network descriptor for the HIP engine; pass it as network= to trainers.ConstrainedAAE')


constrained_adversarial_autoencoder.arch = 'ConstrainedAAE'
constrained_adversarial_autoencoder.output_keys = ('z_', 'x_hat', 'z_rec', 'd_', 'd', 'z_hat', 'd_hat')
