"""Weight initialisation of the reference graph: keras 'glorot_normal' kernels, zero biases, BN gamma=1/beta=0
(multiscale_vae.py:61, layer_blocks.py:15).  Host-side numpy; the arena upload is Engine.set_params."""
from collections import OrderedDict

import numpy as np

_TRUNC_STD = 0.87962566103423978   # std of a unit normal truncated to [-2, 2] (keras VarianceScaling)


def _fans(name, shape):
    if len(shape) == 2:                       # Dense kernel (in, out)
        return shape[0], shape[1]
    kh, kw, a, b = shape
    if name.endswith(".dw.w"):                # DepthwiseConv2D kernel (kh,kw,C,1): keras fans on the full shape
        return kh * kw * a, kh * kw * b
    return kh * kw * a, kh * kw * b           # Conv2D HWIO and Conv2DTranspose (kh,kw,out,in) alike


def truncated_normal(rng, shape, std):
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():                          # resample the tails like tf.random.truncated_normal
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def init_params(param_table, seed=42):
    """param_table: name -> dict(shape=...) in arena order.  Returns name -> float32 array."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, meta in param_table.items():
        shape = tuple(meta["shape"])
        if name.endswith(".w"):
            fi, fo = _fans(name, shape)
            std = np.sqrt(2.0 / (fi + fo)) / _TRUNC_STD
            out[name] = truncated_normal(rng, shape, std)
        elif name.endswith(".gamma"):
            out[name] = np.ones(shape, np.float32)
        else:                                 # biases, BN beta
            out[name] = np.zeros(shape, np.float32)
    return out


def init_state(state_table):
    out = OrderedDict()
    for name, meta in state_table.items():
        out[name] = (np.ones if name.endswith(".var") else np.zeros)(tuple(meta["shape"]), np.float32)
    return out
