"""mobilenetV2_block and resnet_block (reference mvae/layer_blocks.py:468-550, 789-887).
CPU: the oracle restatements replay the reference's own fixtures for these blocks (tests/test_layer_blocks.py:45-52 and
:97-104 -- shape-only tests on zeros: (3,256,256,3) -> 32 channels for resnet_block(x, 32), 3 channels for
mobilenetV2_block(x, 32)) and the reference's argument checks.  GPU: the HIP layers (C ABI mvae_mnv2_* / mvae_resnet_*)
against the oracle's float64 autograd, forward and backward."""
import numpy as np
import pytest

from tests.common import rel_err


def _params(kind, c, f, k=(3, 3), seed=0):
    from oracle.mvae_oracle import block_param_shapes_v2
    rng = np.random.default_rng(seed)
    P = {}
    for name, shp in block_param_shapes_v2(kind, c, f, k).items():
        scale = 0.3 if name.endswith(".w") else 0.1
        P[name] = (scale * rng.standard_normal(shp)).astype(np.float32)
    return P


def test_reference_shape_fixtures_and_checks():
    import torch
    from oracle.mvae_oracle import block_param_shapes_v2, mobilenetV2_block_t, resnet_block_t
    x = torch.zeros((3, 3, 64, 64), dtype=torch.float64)          # the reference uses 256x256; the shape logic is size-free
    T = {k: torch.zeros(s, dtype=torch.float64) for k, s in block_param_shapes_v2("resnet", 3, 32).items()}
    assert "skip.w" in T                                          # 3 != 32 channels: the 1x1 skip convolution (:858-872)
    assert tuple(resnet_block_t(x, T).shape) == (3, 32, 64, 64)   # tests/test_layer_blocks.py:45-52
    T = {k: torch.zeros(s, dtype=torch.float64) for k, s in block_param_shapes_v2("mnv2", 3, 32).items()}
    y = mobilenetV2_block_t(x, T)
    assert tuple(y.shape) == (3, 3, 64, 64) and float(y.abs().max()) == 0.0     # :97-104
    assert "skip.w" not in block_param_shapes_v2("resnet", 32, 32)              # identity skip (:859-862)
    with pytest.raises(ValueError, match="Filters should be > 0"):
        block_param_shapes_v2("mnv2", 3, 0)
    from multiscale_variational_autoencoder_amd.layer_blocks import mobilenetV2_block, resnet_block
    with pytest.raises(ValueError, match="Filters should be > 0"):
        resnet_block((8, 8, 3), 0)
    with pytest.raises(ValueError, match="input_layer cannot be empty"):
        mobilenetV2_block(None, 32)
    with pytest.raises(ValueError, match="Dropout ration"):
        resnet_block((8, 8, 3), 32, dropout_ratio=1.5)
    import mvae.layer_blocks as alias
    assert alias.mobilenetV2_block is mobilenetV2_block and alias.resnet_block is resnet_block


@pytest.mark.gpu
@pytest.mark.parametrize("c,f,hw", [(3, 32, (16, 16)), (32, 64, (12, 20)), (64, 64, (32, 32))])
def test_mobilenetV2_block_forward_backward(c, f, hw):
    from oracle.mvae_oracle import block_forward_backward
    from multiscale_variational_autoencoder_amd.layer_blocks import mobilenetV2_block
    rng = np.random.default_rng(1)
    B = 5
    x = rng.standard_normal((B,) + hw + (c,)).astype(np.float32)
    dy = rng.standard_normal((B,) + hw + (c,)).astype(np.float32)
    P = _params("mnv2", c, f)
    layer = mobilenetV2_block(hw + (c,), f)
    assert set(layer.get_weights()) == set(P)
    layer.set_weights(P)
    y = layer(x)
    dx, G = layer.backward(dy)
    yr, dxr, Gr = block_forward_backward("mnv2", x, P, dy)
    assert y.shape == x.shape and rel_err(y, yr) <= 2e-6
    assert rel_err(dx, dxr) <= 1e-5
    for k in Gr:
        assert rel_err(G[k], Gr[k]) <= 2e-5, (k, rel_err(G[k], Gr[k]))


@pytest.mark.gpu
@pytest.mark.parametrize("c,f,k,act,hw", [(3, 32, (3, 3), "relu", (16, 16)), (32, 32, (3, 3), "relu", (12, 20)),
                                          (32, 64, (5, 5), "linear", (16, 16)), (64, 64, (1, 1), "relu", (32, 32))])
def test_resnet_block_forward_backward(c, f, k, act, hw):
    from oracle.mvae_oracle import block_forward_backward
    from multiscale_variational_autoencoder_amd.layer_blocks import resnet_block
    rng = np.random.default_rng(2)
    B = 4
    x = rng.standard_normal((B,) + hw + (c,)).astype(np.float32)
    dy = rng.standard_normal((B,) + hw + (f,)).astype(np.float32)
    P = _params("resnet", c, f, k)
    layer = resnet_block(hw + (c,), f, kernel_size=k, activation=act)
    assert set(layer.get_weights()) == set(P)
    layer.set_weights(P)
    y = layer(x)
    assert y.shape == (B,) + hw + (f,)
    dx, G = layer.backward(dy)
    yr, dxr, Gr = block_forward_backward("resnet", x, P, dy, activation=act)
    assert rel_err(y, yr) <= 2e-6
    assert rel_err(dx, dxr) <= 1e-5
    for kk in Gr:
        assert rel_err(G[kk], Gr[kk]) <= 2e-5, (kk, rel_err(G[kk], Gr[kk]))
