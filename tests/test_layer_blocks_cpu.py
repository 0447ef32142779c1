"""CPU: the oracle restatement of the remaining block library (oracle/blocks_oracle.py) against everything the reference
pins for it -- the output shapes of tests/test_layer_blocks.py:42-79 (resnet_block (3,256,256,3) -> (3,256,256,32),
attention_block -> (3,256,256,32), self_attention_block -> (3,256,256,3)), replayed at 32x32 as well -- plus internal
consistency (the attention core as the reference writes it, MaxPooling SAME, the facade's shape tables and errors).
"Parity unpinned" for values: the reference holds no value fixture for these blocks."""
import numpy as np
import pytest
import torch

from oracle import blocks_oracle as bo


def _params(kind, c, f, k=(3, 3), seed=0, **kw):
    rng = np.random.default_rng(seed)
    P = {}
    for name, shp in bo.layer_param_shapes(kind, c, f, k, **kw).items():
        if name.endswith(".var"):
            P[name] = 1.0 + 0.3 * rng.uniform(size=shp)
        elif name.endswith(".gamma"):
            P[name] = 1.0 + 0.1 * rng.standard_normal(shp)
        else:
            P[name] = 0.3 * rng.standard_normal(shp)
    return P


@pytest.mark.parametrize("kind,filters,out_c", [("resnet", 32, 32), ("attention", 32, 32), ("self_attention", 32, 3)])
def test_reference_shape_fixtures(kind, filters, out_c):          # tests/test_layer_blocks.py:42-79
    for hw in (256, 32):
        b = 3 if hw == 32 else 1
        x = np.zeros((b, hw, hw, 3))
        k = (3, 3) if kind == "resnet" else (1, 1)
        y, dx, G = bo.layer_forward_backward(kind, x, _params(kind, 3, filters, k), np.zeros(bo.layer_output_shape(kind, x.shape, filters)))
        assert y.shape == (b, hw, hw, out_c) == bo.layer_output_shape(kind, x.shape, filters)
        assert dx.shape == x.shape


@pytest.mark.parametrize("kind,kw,shape", [("spatial_mask", {}, (2, 8, 12, 5)), ("spatial_mask", {"flatten": True}, (2, 8, 12, 1)),
                                           ("channel_mask", {}, (2, 5)), ("channel_mask", {"shared": False}, (2, 5)),
                                           ("excite_inhibit", {}, (2, 8, 12, 5)),
                                           ("resnet", {"strides": (2, 2)}, (2, 4, 6, 7)), ("resnet", {"strides": (2, 3), "use_batchnorm": True}, (2, 4, 4, 7)),
                                           ("mnv2", {"use_batchnorm": True}, (2, 8, 12, 5))])
def test_sibling_shapes_and_mask_range(kind, kw, shape):
    x = np.random.default_rng(1).standard_normal((2, 8, 12, 5))
    P = _params(kind, 5, 7, **{k: v for k, v in kw.items() if k in ("flatten", "shared", "use_batchnorm")})
    y, dx, G = bo.layer_forward_backward(kind, x, P, np.ones(shape), **kw)
    assert y.shape == shape and dx.shape == x.shape
    assert y.shape == bo.layer_output_shape(kind, x.shape, 7, kw.get("strides", (1, 1)), **{k: v for k, v in kw.items() if k == "flatten"})
    if kind.endswith("mask"):
        assert (y > 0).all() and (y < 1).all()                    # attenuate_activation maps into (0, 1)
    assert set(G) == {k for k in P if not k.endswith((".mean", ".var"))}


def test_attention_core_is_the_keras_dot_reshape_sequence():
    """The einsum form of oracle.attention_block_t against a literal replay of the Keras calls on NumPy:
    Reshape((HW, F)), Permute((2, 1)), Dot(axes=(1, 2)), Softmax(), Dot(axes=(1, 2)), Reshape((H, W, F))."""
    rng = np.random.default_rng(3)
    B, H, W, Fn = 2, 3, 5, 4
    th, ph, g = (rng.standard_normal((B, H, W, Fn)) for _ in range(3))
    thf, phf, gf = th.reshape(B, H * W, Fn), np.transpose(ph.reshape(B, H * W, Fn), (0, 2, 1)), g.reshape(B, H * W, Fn)
    S = np.einsum("bpi,bjp->bij", thf, phf)                       # batch_dot contracting thf axis 1 with phf axis 2
    A = np.exp(S - S.max(-1, keepdims=True)); A /= A.sum(-1, keepdims=True)
    O = np.einsum("bij,bpi->bjp", A, gf)                          # contracting A axis 1 with gf axis 2 -> (B, F, HW)
    want = O.reshape(B, H, W, Fn)
    T = {}
    for n in ("theta", "phi", "g"):                              # identity 1x1 convolutions: the block's input IS theta = phi = g
        T[n + ".w"] = torch.eye(Fn, dtype=torch.float64).view(1, 1, Fn, Fn); T[n + ".b"] = torch.zeros(Fn, dtype=torch.float64)
    x = torch.as_tensor(th).permute(0, 3, 1, 2)
    got = bo.attention_block_t(x, T).permute(0, 2, 3, 1).numpy()
    S1 = np.einsum("bpi,bpj->bij", thf, thf); A1 = np.exp(S1 - S1.max(-1, keepdims=True)); A1 /= A1.sum(-1, keepdims=True)
    want1 = np.einsum("bij,bpi->bjp", A1, thf).reshape(B, H, W, Fn)
    assert np.abs(got - want1).max() < 1e-12
    assert want.shape == (B, H, W, Fn)


def test_maxpool_same_matches_a_loop():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 2, 7, 5))
    for pool, st in (((3, 3), (2, 2)), ((3, 4), (2, 3))):
        y = bo.maxpool_same(torch.as_tensor(x), pool, st).numpy()
        oh, ow = -(-7 // st[0]), -(-5 // st[1])
        pt = max((oh - 1) * st[0] + pool[0] - 7, 0) // 2
        pl = max((ow - 1) * st[1] + pool[1] - 5, 0) // 2
        for oy in range(oh):
            for ox in range(ow):
                ys = [yy for yy in range(oy * st[0] - pt, oy * st[0] - pt + pool[0]) if 0 <= yy < 7]
                xs = [xx for xx in range(ox * st[1] - pl, ox * st[1] - pl + pool[1]) if 0 <= xx < 5]
                assert np.allclose(y[0, :, oy, ox], x[0][:, ys][:, :, xs].max(axis=(1, 2)))


def test_facade_errors_and_tables(hip_lib):
    """The reference's ValueErrors (layer_blocks.py:216-223, 675-682, 818-823) and the facade's weight tables."""
    import mvae.layer_blocks as lb
    from multiscale_variational_autoencoder_amd.layer_ops import _shapes
    with pytest.raises(ValueError, match="input_layer cannot be empty"):
        lb.attention_block(None)
    with pytest.raises(ValueError, match="4d tensors"):
        lb.attention_block((8, 8))
    with pytest.raises(ValueError, match="Filters should be > 0"):
        lb.self_attention_block((8, 8, 3), 0)
    with pytest.raises(ValueError, match="works only on 4d tensors"):
        lb.excite_inhibit_block((8, 3))
    with pytest.raises(ValueError, match="input_layer cannot be empty"):
        lb.excite_inhibit_spatial_mask_block(None)
    blk = lb.resnet_block((8, 8, 3), 16, strides=(2, 2), use_batchnorm=True)
    assert type(blk).__name__ == "ResnetBlockGeneral"
    assert list(blk.get_weights()) == list(bo.layer_param_shapes("resnet", 3, 16, (3, 3), use_batchnorm=True))
    for kind, kw in (("attention", {}), ("self_attention", {}), ("spatial_mask", {"flatten": True}), ("channel_mask", {"shared": False}),
                     ("excite_inhibit", {}), ("mnv2", {"use_batchnorm": True})):
        assert _shapes(kind, 3, 32, (3, 3), **kw) == bo.layer_param_shapes(kind, 3, 32, (3, 3), **kw)
    w = lb.excite_inhibit_block((8, 8, 3), 8).get_weights()
    assert w["cmask.de.w"].shape == (8, 3) and w["smask.e1.w"].shape == (1, 1, 8, 3) and np.all(w["conv0.b"] == 0)
