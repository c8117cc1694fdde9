"""trainers/AE.py — dense AE: loss = mean_n sum_hwc |x_hat - x| (AE.py:28-29); dropout only on z (autoencoder.py:29-30).  The same
trainer drives models/autoencoder_spatial.py (latent = the encoder feature map, dropout on it)."""
import numpy as np

from .AEMODEL import AEMODEL, Phase, indicate_early_stopping  # noqa: F401


class AE(AEMODEL):
    ARCH = 'AE'
    ARCHS = ('AE', 'AE_spatial')
    SCALAR_KEYS = ('reconstructionLoss', 'loss')

    def _noise_layout(self, dropout):
        if not dropout or self.config.dropout_rate <= 0:
            return []
        e = self.engine
        return [('z', (e.inter, e.inter, e._cenc()) if self.arch == 'AE_spatial' else self.config.zDim, 'keep')]      # autoencoder.py:29-30

    def _draw(self, n, dropout):
        if not dropout or self.config.dropout_rate <= 0:
            return None, None
        r = float(self.config.dropout_rate)
        if self.arch == 'AE_spatial':
            shape = (n, self.engine.inter, self.engine.inter, self.engine._cenc())
        else:
            shape = (n, self.config.zDim)
        keep = (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)
        return None, {'z': keep}
