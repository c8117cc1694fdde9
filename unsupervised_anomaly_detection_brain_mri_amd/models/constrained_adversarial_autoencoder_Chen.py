"""models/constrained_adversarial_autoencoder_Chen.py:11-162 -- network descriptor: encoder / decoder of pre-activation residual blocks (LayerNorm over
(H, W), k3 convolutions, AvgPool(1x1 conv) / k1 s2 transposed-conv shortcuts; base width 64), Dense latent, MLP critic 400-200-1 on z with one
scalar interpolation coefficient per run.  The graph itself lives in csrc/uad_gan.hip (uad_gan_create with UAD_GAN_AAE, aae_kind 7); trained
by trainers/ConstrainedAAE.py; intermediateResolutions = height / 8."""


def constrained_adversarial_autoencoder_Chen(*args, **kw):
    raise RuntimeError('constrained_adversarial_autoencoder_Chen() is a network descriptor for the HIP engine; pass it as network= to trainers.ConstrainedAAE')


constrained_adversarial_autoencoder_Chen.arch = 'CAAE_Chen'
constrained_adversarial_autoencoder_Chen.output_keys = ('z_', 'x_hat', 'z_rec', 'd_', 'd', 'z_hat', 'd_hat')
