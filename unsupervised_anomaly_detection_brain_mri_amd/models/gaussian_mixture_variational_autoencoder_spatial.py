"""models/gaussian_mixture_variational_autoencoder_spatial.py:9-65 — spatial GMVAE: unified encoder -> 1x1 heads q(w|x),
q(z|x) on the r x r map, p(z|w,c) mixture heads, softmax mixture posterior p(c); the unified decoder is applied to the
ENCODER FEATURE MAP (:52-56), not to z.  The graph itself lives in csrc/uad_model.hip + csrc/uad_gmvae.hip
(uad_create with UAD_ARCH_GMVAE_SPATIAL)."""


def gaussian_mixture_variational_autoencoder_spatial(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('gaussian_mixture_variational_autoencoder_spatial() is a network descriptor for the HIP engine; '
                       'pass it as network= to a trainer')


gaussian_mixture_variational_autoencoder_spatial.arch = 'GMVAE_spatial'
gaussian_mixture_variational_autoencoder_spatial.output_keys = (
    'w_mu', 'w_log_sigma', 'w_sampled', 'z_mu', 'z_log_sigma', 'z_sampled', 'z_wc_mus', 'z_wc_log_sigma_invs',
    'z_wc_sampled', 'xz_mu', 'pc_logit', 'pc')
