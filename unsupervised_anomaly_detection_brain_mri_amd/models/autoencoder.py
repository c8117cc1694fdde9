"""models/autoencoder.py:9-40 — dense-bottleneck AE on the unified encoder/decoder (models/customlayers.py:16-38).
The graph itself lives in csrc/uad_model.hip (uad_create with UAD_ARCH_AE)."""


def autoencoder(x=None, dropout_rate=None, dropout=None, config=None):
    raise RuntimeError('autoencoder() is a network descriptor for the HIP engine; pass it as network= to a trainer')


autoencoder.arch = 'AE'
autoencoder.output_keys = ('z', 'x_hat')           # models/autoencoder.py:29,38
