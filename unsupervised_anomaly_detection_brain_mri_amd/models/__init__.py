"""Network descriptors with the reference's model-function names.  In the reference a model is a function that builds
a TF graph (models/<name>.py::<name>); here it is a marker whose `__name__` selects the HIP engine's architecture
(the trainer reads `network.__name__` for paths exactly like trainers/AEMODEL.py:32-35 / utils/Evaluation.py:382)."""
from .autoencoder import autoencoder  # noqa: F401
from .variational_autoencoder import variational_autoencoder  # noqa: F401
from .context_encoder_variational_autoencoder import context_encoder_variational_autoencoder  # noqa: F401
from .gaussian_mixture_variational_autoencoder_spatial import gaussian_mixture_variational_autoencoder_spatial  # noqa: F401
from .fanogan import fanogan  # noqa: F401
from .fanogan_schlegl import fanogan_schlegl  # noqa: F401
from .autoencoder_spatial import autoencoder_spatial  # noqa: F401
from .anovaegan import anovaegan  # noqa: F401
from .constrained_autoencoder import constrained_autoencoder  # noqa: F401
from .adversarial_autoencoder import adversarial_autoencoder  # noqa: F401
from .constrained_adversarial_autoencoder import constrained_adversarial_autoencoder  # noqa: F401
from .gaussian_mixture_variational_autoencoder import gaussian_mixture_variational_autoencoder  # noqa: F401
from .variational_autoencoder_Zimmerer import variational_autoencoder_Zimmerer  # noqa: F401
from .context_encoder_variational_autoencoder_Zimmerer import context_encoder_variational_autoencoder_Zimmerer  # noqa: F401
from .gaussian_mixture_variational_autoencoder_You import gaussian_mixture_variational_autoencoder_You  # noqa: F401
from .constrained_adversarial_autoencoder_Chen import constrained_adversarial_autoencoder_Chen  # noqa: F401
