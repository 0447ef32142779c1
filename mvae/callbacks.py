"""`from mvae.callbacks import SaveIntermediateResultsCallback` (reference mvae/callbacks.py:16)."""
from multiscale_variational_autoencoder_amd.callbacks import SaveIntermediateResultsCallback, collage  # noqa: F401
