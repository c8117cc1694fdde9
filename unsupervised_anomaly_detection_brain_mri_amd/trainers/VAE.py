"""trainers/VAE.py — VAE: rec = sum_hwc |x_hat - x|, kl = 0.5 sum(mu^2 + sigma^2 - log sigma^2 - 1),
loss = mean(rec + kl) (VAE.py:36-42); dropout on mu, log_sigma and dec_dense (variational_autoencoder.py:31-35)."""
import numpy as np

from .AEMODEL import AEMODEL, Phase, indicate_early_stopping  # noqa: F401


class VAE(AEMODEL):
    class Config(AEMODEL.Config):
        def __init__(self):
            super().__init__('VAE')

    ARCH = 'VAE'
    ARCHS = ('VAE', 'VAE_Zimmerer')          # models/variational_autoencoder.py | models/variational_autoencoder_Zimmerer.py (no dropout layers)
    SCALAR_KEYS = ('reconstructionLoss', 'kl', 'loss')

    def _make_engine(self, device):
        if self.arch != 'VAE_Zimmerer':
            return super()._make_engine(device)
        from ..gan_engine import ZimmererEngine
        c = self.config
        return ZimmererEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim,
                              max_batch=max(int(c.batchsize), 1), device=device)

    def _noise_layout(self, dropout):
        z = self.config.zDim
        lay = [('eps', z, 'normal')]
        if dropout and self.config.dropout_rate > 0 and self.arch != 'VAE_Zimmerer':
            lay += [('mu', z, 'keep'), ('sigma', z, 'keep'), ('dec', self.engine.flat, 'keep')]      # variational_autoencoder.py:31-35
        return lay

    def _draw(self, n, dropout):
        z = self.config.zDim
        eps = self.rng.standard_normal((n, z)).astype(np.float32)
        if not dropout or self.config.dropout_rate <= 0 or self.arch == 'VAE_Zimmerer':
            return eps, None
        r = float(self.config.dropout_rate)
        keep = lambda shape: (self.rng.random(shape) >= r).astype(np.float32) / (1.0 - r)
        return eps, {'mu': keep((n, z)), 'sigma': keep((n, z)), 'dec': keep((n, self.engine.flat))}
