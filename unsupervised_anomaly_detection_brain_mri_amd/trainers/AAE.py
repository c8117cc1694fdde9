"""trainers/AAE.py — adversarial autoencoder (see trainers/ConstrainedAE.py for the shared implementation)."""
from .AEMODEL import AEMODEL
from .ConstrainedAE import _LatentAE


class AAE(_LatentAE):
    class Config(AEMODEL.Config):
        def __init__(self):          # trainers/AAE.py:11-14
            super().__init__('AAE')
            self.scale = 10.0

    ARCH = 'AAE'
    KIND = 'aae'
