"""`mvae.layer_blocks` of the MI355X build: the stand-alone Laplacian pyramid (reference mvae/layer_blocks.py:23-185) and the
mobilenetV2 / resnet blocks (:468-550, :789-887)."""
from multiscale_variational_autoencoder_amd.layer_blocks import (  # noqa: F401
    DEFAULT_GAUSSIAN_KERNEL_SIZE, DEFAULT_GAUSSIAN_XY_MAX, gaussian_kernel, laplacian_transform_merge,
    laplacian_transform_split, mobilenetV2_block, resnet_block)
