"""models/variational_autoencoder_Zimmerer.py:7-32 -- network descriptor: k4 s2 convolutions 16-64-256-1024 with tf.nn.leaky_relu, Dense mu /
log-sigma heads on the flattened map, Dense back to [r, r, 1024], four k4 s2 transposed convolutions, a k4 convolution to one channel; no
normalisation, no dropout.  The graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE, aae_kind 4); trained by trainers/VAE.py."""


def variational_autoencoder_Zimmerer(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('variational_autoencoder_Zimmerer() is a network descriptor for the HIP engine; pass it as network= to a trainer')


variational_autoencoder_Zimmerer.arch = 'VAE_Zimmerer'
variational_autoencoder_Zimmerer.output_keys = ('z_mu', 'z_log_sigma', 'z_sigma', 'x_hat')
