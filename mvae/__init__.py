"""`from mvae import MultiscaleVAE` keeps working (reference mvae/__init__.py:9): thin alias of the MI355X build."""
from multiscale_variational_autoencoder_amd import MultiscaleVAE, step_decay_schedule  # noqa: F401
from multiscale_variational_autoencoder_amd import schedule  # noqa: F401
from . import callbacks  # noqa: F401
from . import layer_blocks  # noqa: F401

__all__ = ["MultiscaleVAE", "schedule", "callbacks", "layer_blocks", "step_decay_schedule"]
