"""GPU: the remaining block library (attention_block, self_attention_block, the excite / inhibit masks and block,
resnet_block with strides / BatchNormalization, mobilenetV2_block with BatchNormalization; reference
mvae/layer_blocks.py:191-412, 468-550, 654-887) assembled from the HIP layer operators, against the float64 oracle
(oracle/blocks_oracle.py).  Tolerances (float32 kernels vs float64): outputs <= 2e-6 relative (norm-wise), input gradient
<= 2e-5, weight gradients <= 2e-5 per tensor (the bars of tests/test_blocks_v2_resnet.py; bias / BatchNorm vectors 4x that); blocks with BatchNormalization
over a small batch or max pooling get 1e-4 (normalisation amplifies rounding; documented per case)."""
import numpy as np
import pytest

from oracle import blocks_oracle as bo
from tests.common import rel_err

pytestmark = pytest.mark.gpu


def _randomise(blk, seed):
    rng = np.random.default_rng(seed)
    w = blk.get_weights()
    for k, v in w.items():
        if k.endswith(".var"):
            w[k] = (1.0 + 0.3 * rng.uniform(size=v.shape)).astype(np.float32)
        elif k.endswith(".gamma"):
            w[k] = (1.0 + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        elif k.endswith((".b", ".beta", ".mean")):
            w[k] = (0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        else:
            w[k] = (v + 0.05 * rng.standard_normal(v.shape)).astype(np.float32)
    blk.set_weights(w)
    return w


def _check(kind, blk, x, out_shape, tol=(2e-6, 2e-5, 2e-5), training=True, **kw):
    w = _randomise(blk, 7)
    y = blk.forward(x, training=training)
    dy = np.random.default_rng(9).standard_normal(out_shape).astype(np.float32)
    dx, G = blk.backward(dy)
    yr, dxr, Gr = bo.layer_forward_backward(kind, x, w, dy, training=training, **kw)
    assert y.shape == tuple(out_shape) == yr.shape
    assert rel_err(y, yr) <= tol[0], ("y", rel_err(y, yr))
    assert rel_err(dx, dxr) <= tol[1], ("dx", rel_err(dx, dxr))
    gscale = max(np.linalg.norm(g) for g in Gr.values())
    for k, g in Gr.items():
        err = np.linalg.norm((G[k].astype(np.float64) - g).ravel()) / max(np.linalg.norm(g.ravel()), 1e-3 * gscale)
        # bias / BatchNorm vectors are plain column sums over all pixels of terms of both signs: 4x the weights' bar
        assert err <= tol[2] * (4 if g.ndim == 1 else 1), (k, err)
    return y


@pytest.mark.parametrize("shape,filters,k,act", [((2, 16, 16, 3), 32, (1, 1), "linear"), ((3, 12, 20, 8), 16, (3, 3), "relu"),
                                                 ((1, 64, 64, 3), 64, (1, 1), "tanh")])
def test_attention_block(shape, filters, k, act):
    import mvae.layer_blocks as lb
    x = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
    blk = lb.attention_block(shape[1:], filters, k, act)
    # the F x F scores are sums over ALL pixels (up to 4096 here) whose float32 rounding error, of the size of |S| * 1e-7,
    # goes through exp(): the softmax carries a relative error of that ABSOLUTE size (|S| reaches tens): 1e-5 / 5e-5
    _check("attention", blk, x, shape[:3] + (filters,), (1e-5, 5e-5, 5e-5), activation=act)


@pytest.mark.parametrize("shape,filters,act", [((2, 16, 16, 3), 32, "linear"), ((2, 8, 24, 16), 8, "relu")])
def test_self_attention_block(shape, filters, act):
    import mvae.layer_blocks as lb
    x = np.random.default_rng(2).standard_normal(shape).astype(np.float32)
    blk = lb.self_attention_block(shape[1:], filters, (1, 1), act)
    _check("self_attention", blk, x, shape, (1e-5, 5e-5, 5e-5), activation=act)


def test_reference_shape_fixture_attention_256():
    """tests/test_layer_blocks.py:58-79 at their own size: (3, 256, 256, 3) zeros -> (3, 256, 256, 32) / (3, 256, 256, 3)."""
    import mvae.layer_blocks as lb
    x = np.zeros((3, 256, 256, 3), np.float32)
    assert lb.attention_block((256, 256, 3), 32).forward(x).shape == (3, 256, 256, 32)
    y = lb.self_attention_block((256, 256, 3), 32).forward(x)
    assert y.shape == (3, 256, 256, 3) and np.isfinite(y).all()
    assert lb.resnet_block((256, 256, 3), 32).forward(x).shape == (3, 256, 256, 32)      # :42-50


@pytest.mark.parametrize("flatten", [False, True])
def test_spatial_mask(flatten):
    import mvae.layer_blocks as lb
    shape = (2, 12, 20, 6)
    x = np.random.default_rng(3).standard_normal(shape).astype(np.float32)
    blk = lb.excite_inhibit_spatial_mask_block(shape[1:], 16, (3, 3), flatten=flatten)
    y = _check("spatial_mask", blk, x, shape[:3] + ((1,) if flatten else (6,)), flatten=flatten)
    assert (y > 0).all() and (y < 1).all()
    assert np.allclose(lb.attenuate_activation(np.array([0.0, 1.0, -1.0], np.float32)), (np.tanh([0.0, 4.0, -4.0]) + 1) / 2, atol=1e-6)


@pytest.mark.parametrize("shared", [True, False])
def test_channel_mask(shared):
    import mvae.layer_blocks as lb
    shape = (3, 12, 20, 6)
    x = np.random.default_rng(4).standard_normal(shape).astype(np.float32)
    blk = lb.excite_inhibit_channel_mask_block(shape[1:], 16, (3, 3), shared=shared)
    _check("channel_mask", blk, x, (3, 6), shared=shared)


def test_excite_inhibit_block():
    import mvae.layer_blocks as lb
    shape = (2, 16, 16, 8)
    x = np.random.default_rng(5).standard_normal(shape).astype(np.float32)
    _check("excite_inhibit", lb.excite_inhibit_block(shape[1:], 16, (3, 3)), x, shape)


@pytest.mark.parametrize("shape,filters,strides,bn,act", [((2, 16, 16, 8), 16, (2, 2), False, "relu"), ((2, 15, 17, 8), 8, (2, 2), False, "relu"),
                                                           ((4, 16, 16, 8), 16, (1, 1), True, "relu"), ((4, 12, 20, 6), 16, (2, 3), True, "linear")])
def test_resnet_block_strides_and_batchnorm(shape, filters, strides, bn, act):
    import mvae.layer_blocks as lb
    x = np.random.default_rng(6).standard_normal(shape).astype(np.float32)
    blk = lb.resnet_block(shape[1:], filters, (3, 3), strides, act, use_batchnorm=bn)
    out = (shape[0], -(-shape[1] // strides[0]), -(-shape[2] // strides[1]), filters)
    # BatchNormalization over a few hundred rows: the normalisation's 1 / sqrt(var) amplifies float32 rounding
    tol = (2e-6, 1e-4, 1e-4) if bn else (2e-6, 2e-5, 2e-5)
    _check("resnet", blk, x, out, tol, training=True, strides=strides, use_batchnorm=bn, activation=act)
    if bn:                                                                   # inference mode: the moving statistics
        _check("resnet", blk, x, out, tol, training=False, strides=strides, use_batchnorm=bn, activation=act)


def test_mobilenetV2_block_batchnorm():
    import mvae.layer_blocks as lb
    shape = (4, 16, 16, 8)
    x = np.random.default_rng(8).standard_normal(shape).astype(np.float32)
    blk = lb.mobilenetV2_block(shape[1:], 32, use_batchnorm=True)
    assert type(blk).__name__ == "MobileNetV2BlockBN"
    _check("mnv2", blk, x, shape, (2e-6, 1e-4, 1e-4), training=True, use_batchnorm=True)
    _check("mnv2", blk, x, shape, (2e-6, 1e-4, 1e-4), training=False, use_batchnorm=True)
