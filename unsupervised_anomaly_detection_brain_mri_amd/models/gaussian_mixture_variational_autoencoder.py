"""models/gaussian_mixture_variational_autoencoder.py:11-76 — dense GMVAE: unified encoder -> 1x1 conv C/8 -> flatten -> Dense heads
q(w|x), q(z|x); p(z|w,c) = two Dense layers on w_sampled; softmax mixture posterior p(c|z); the decoder consumes
dropout(Dense(z_sampled)) through the reverse 1x1 conv.  The graph itself lives in csrc/uad_gan.hip + csrc/uad_gmd.hip
(uad_gan_create with UAD_GAN_AAE, aae_kind 3)."""


def gaussian_mixture_variational_autoencoder(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('gaussian_mixture_variational_autoencoder() is a network descriptor for the HIP engine; pass it as network= to a trainer')


gaussian_mixture_variational_autoencoder.arch = 'GMVAE'
gaussian_mixture_variational_autoencoder.output_keys = (
    'w_mu', 'w_log_sigma', 'w_sampled', 'z_mu', 'z_log_sigma', 'z_sampled', 'z_wc_mus', 'z_wc_log_sigma_invs', 'z_wc_sampled', 'xz_mu',
    'pc_logit', 'pc')
