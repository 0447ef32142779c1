"""`mvae.layer_blocks` of the MI355X build: the stand-alone Laplacian pyramid (reference mvae/layer_blocks.py:23-185), the
mobilenetV2 / resnet blocks (:468-550, :789-887), the attention blocks (:654-783) and the excite / inhibit masks (:191-412)."""
from multiscale_variational_autoencoder_amd.layer_blocks import (  # noqa: F401
    DEFAULT_ATTENUATION_MULTIPLIER, DEFAULT_GAUSSIAN_KERNEL_SIZE, DEFAULT_GAUSSIAN_XY_MAX, attention_block,
    attenuate_activation, excite_inhibit_block, excite_inhibit_channel_mask_block, excite_inhibit_spatial_mask_block,
    gaussian_kernel, laplacian_transform_merge, laplacian_transform_split, mobilenetV2_block, resnet_block,
    self_attention_block)
