"""models/variational_autoencoder.py:9-47 — VAE with mu / log-sigma heads, z = mu + eps * exp(log_sigma).
The graph itself lives in csrc/uad_model.hip (uad_create with UAD_ARCH_VAE)."""


def variational_autoencoder(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('variational_autoencoder() is a network descriptor for the HIP engine; pass it as network= to a trainer')


variational_autoencoder.arch = 'VAE'
variational_autoencoder.output_keys = ('z_mu', 'z_log_sigma', 'z_sigma', 'x_hat')   # variational_autoencoder.py:31-33,45
