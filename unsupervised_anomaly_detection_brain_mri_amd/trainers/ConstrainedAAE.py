"""trainers/ConstrainedAAE.py — constrained adversarial autoencoder (see trainers/ConstrainedAE.py for the shared implementation)."""
from .AEMODEL import AEMODEL
from .ConstrainedAE import _LatentAE


class ConstrainedAAE(_LatentAE):
    class Config(AEMODEL.Config):
        def __init__(self):          # trainers/ConstrainedAAE.py:11-15
            super().__init__('ConstrainedAAE')
            self.rho = 1
            self.scale = 10.0

    ARCH = 'ConstrainedAAE'
    ARCHS = ('ConstrainedAAE', 'CAAE_Chen')      # models/constrained_adversarial_autoencoder.py | ..._Chen.py (residual blocks, no dropout, scalar eps)
    KIND = 'constrained_aae'

    def _make_engine(self, device):
        if self.arch != 'CAAE_Chen':
            return super()._make_engine(device)
        from ..gan_engine import GanEngine
        c = self.config
        self.KIND = 'caae_chen'
        return GanEngine(c.outputHeight, c.outputWidth, c.numChannels, int(c.intermediateResolutions[0]), c.zDim, max_batch=max(int(c.batchsize), 1),
                         scale=float(getattr(c, 'scale', 10.0)), device=device, variant='aae', aae_kind='caae_chen', rho=float(getattr(c, 'rho', 1.0)),
                         dim=64, math='f32')
