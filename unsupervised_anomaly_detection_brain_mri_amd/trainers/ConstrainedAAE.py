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
    KIND = 'constrained_aae'
