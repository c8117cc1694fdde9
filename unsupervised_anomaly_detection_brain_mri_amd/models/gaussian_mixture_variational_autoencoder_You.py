"""models/gaussian_mixture_variational_autoencoder_You.py:8-85 -- network descriptor: the original-architecture spatial GMVAE (six k3 convolutions
of 64 filters with ReLU, 1x1 latent heads on the H/4 map, decoder on z_sampled with nearest-neighbour upsampling).  The graph itself lives in
csrc/uad_gan.hip + csrc/uad_gmvae.hip (uad_gan_create with UAD_GAN_AAE, aae_kind 6); trained by trainers/GMVAE_spatial.py."""


def gaussian_mixture_variational_autoencoder_You(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('gaussian_mixture_variational_autoencoder_You() is a network descriptor for the HIP engine; pass it as network= to a trainer')


gaussian_mixture_variational_autoencoder_You.arch = 'GMVAE_You'
gaussian_mixture_variational_autoencoder_You.output_keys = (
    'w_mu', 'w_log_sigma', 'w_sampled', 'z_mu', 'z_log_sigma', 'z_sampled', 'z_wc_mus', 'z_wc_log_sigma_invs', 'z_wc_sampled', 'xz_mu', 'pc_logit', 'pc')
