"""models/constrained_autoencoder.py:9-48 -- network descriptor; the graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE), trained by
trainers/ConstrainedAE.py."""


def constrained_autoencoder(*args, **kw):
    raise RuntimeError('constrained_autoencoder() is a network descriptor for the HIP engine; pass it as network= to trainers.ConstrainedAE')


constrained_autoencoder.arch = 'ConstrainedAE'
constrained_autoencoder.output_keys = ('z', 'x_hat', 'z_rec')
