"""models/context_encoder_variational_autoencoder_Zimmerer.py:8-45 -- network descriptor: the Zimmerer stack (k4 s2 convolutions 16-64-256-1024,
tf.nn.leaky_relu, Dense heads, k4 transposed convolutions; scopes Encoder / Bottleneck / Decoder) with the shared-weight context branch that
decodes dec_dense(mu_layer(flatten_ce)).  The graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE, aae_kind 5); trained by
trainers/ceVAE.py."""


def context_encoder_variational_autoencoder_Zimmerer(x=None, x_ce=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('context_encoder_variational_autoencoder_Zimmerer() is a network descriptor for the HIP engine; pass it as network= to a trainer')


context_encoder_variational_autoencoder_Zimmerer.arch = 'ceVAE_Zimmerer'
context_encoder_variational_autoencoder_Zimmerer.output_keys = ('z_mu', 'z_log_sigma', 'z_sigma', 'x_hat', 'x_hat_ce')
