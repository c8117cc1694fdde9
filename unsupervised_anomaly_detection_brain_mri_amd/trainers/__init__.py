from .AE import AE  # noqa: F401
from .VAE import VAE  # noqa: F401
from .AEMODEL import Phase  # noqa: F401
from .ceVAE import ceVAE  # noqa: F401
