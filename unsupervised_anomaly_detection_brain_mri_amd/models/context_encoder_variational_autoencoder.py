"""models/context_encoder_variational_autoencoder.py:9-59 — ceVAE: the same encoder / bottleneck / decoder layers are
applied to x (full VAE path) and to the context-masked x_ce (mu only, z_ce = z_mu_ce).  The graph itself lives in
csrc/uad_model.hip (uad_create with UAD_ARCH_CEVAE), which runs both branches as one 2n-sample pass."""


def context_encoder_variational_autoencoder(x=None, x_ce=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('context_encoder_variational_autoencoder() is a network descriptor for the HIP engine; '
                       'pass it as network= to a trainer')


context_encoder_variational_autoencoder.arch = 'ceVAE'
# context_encoder_variational_autoencoder.py:36-39,52,57
context_encoder_variational_autoencoder.output_keys = ('z_mu', 'z_mu_ce', 'z_log_sigma', 'z_sigma', 'x_hat', 'x_hat_ce')
