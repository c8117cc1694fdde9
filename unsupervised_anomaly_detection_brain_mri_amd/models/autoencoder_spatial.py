"""models/autoencoder_spatial.py:7-27 -- spatial AE: unified encoder -> dropout on the feature map (the latent `z`) -> unified decoder;
trained by trainers/AE.py.  The graph itself lives in csrc/uad_model.hip (uad_create with UAD_ARCH_AE_SPATIAL)."""


def autoencoder_spatial(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('autoencoder_spatial() is a network descriptor for the HIP engine; pass it as network= to trainers.AE')


autoencoder_spatial.arch = 'AE_spatial'
autoencoder_spatial.output_keys = ('z', 'x_hat')          # autoencoder_spatial.py:17,25
