"""trainers/AE.py — dense AE: loss = mean_n sum_hwc |x_hat - x| (AE.py:28-29); dropout only on z (autoencoder.py:29-30)."""
import numpy as np

from .AEMODEL import AEMODEL, Phase, indicate_early_stopping  # noqa: F401


class AE(AEMODEL):
    ARCH = 'AE'
    SCALAR_KEYS = ('reconstructionLoss', 'loss')

    def _draw(self, n, dropout):
        if not dropout or self.config.dropout_rate <= 0:
            return None, None
        r = float(self.config.dropout_rate)
        keep = (self.rng.random((n, self.config.zDim)) >= r).astype(np.float32) / (1.0 - r)
        return None, {'z': keep}
