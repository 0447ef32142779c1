"""
multiscale_variational_autoencoder_amd -- the multiscale-VAE train-step hot path of
NikolasMarkou/multiscale_variational_autoencoder as hand-written HIP kernels for AMD Instinct MI355X (gfx950),
behind the reference's own Python surface (`mvae.MultiscaleVAE`, reference mvae/__init__.py:7-15).
"""
from .multiscale_vae import MultiscaleVAE
from .schedule import step_decay_schedule
from .engine import Engine, MvaeError

__all__ = ["MultiscaleVAE", "step_decay_schedule", "Engine", "MvaeError"]
__version__ = "0.1.0"
