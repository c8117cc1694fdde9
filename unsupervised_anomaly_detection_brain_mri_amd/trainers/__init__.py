from .AE import AE  # noqa: F401
from .VAE import VAE  # noqa: F401
from .AEMODEL import Phase  # noqa: F401
from .ceVAE import ceVAE  # noqa: F401
from .GMVAE_spatial import GMVAE_spatial  # noqa: F401
from .fAnoGAN import fAnoGAN  # noqa: F401
from .AnoVAEGAN import AnoVAEGAN  # noqa: F401
from .VAE_You import VAE_You  # noqa: F401
