"""CPU: pin the oracle against every golden vector the reference's own tests hold for this path
(/root/reference/tests/test_layer_blocks.py:9-39: Gaussian blur of zeros / ones, :118-131 pyramid shapes) and
cross-check the torch restatement against the independent numpy loop restatement (oracle/np_ops.py)."""
import numpy as np
import pytest
import torch

from oracle import np_ops
from oracle.mvae_oracle import (Oracle, OracleConfig, conv2d_same, conv2d_transpose_same, gaussian_kernel,
                                param_table)
import torch.nn.functional as F


def _blur_default(x_nhwc):
    """gaussian_filter_block with the reference tests' default xy_max=(1,1) (layer_blocks.py:16)."""
    g = torch.as_tensor(gaussian_kernel((3, 3), (1, 1)))
    x = torch.as_tensor(x_nhwc).permute(0, 3, 1, 2)
    c = x.shape[1]
    w = g.view(1, 1, 3, 3).repeat(c, 1, 1, 1).to(x.dtype)
    return F.conv2d(F.pad(x, (1, 1, 1, 1)), w, None, groups=c).permute(0, 2, 3, 1).numpy()


def test_ref_gaussian_all_zeros():            # tests/test_layer_blocks.py:9-17
    y = _blur_default(np.zeros((3, 256, 256, 3)))
    assert y.shape == (3, 256, 256, 3) and np.all(y == 0.0)


@pytest.mark.parametrize("shape,hi", [((3, 16, 16, 1), 15), ((3, 9, 9, 7), 8)])   # :20-39
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ref_gaussian_all_ones_interior_exact(shape, hi, dtype):
    y = _blur_default(np.ones(shape, dtype))
    assert y.shape == shape
    assert np.all(y[:, 1:hi, 1:hi, :] == 1.0)
    # zero SAME padding: borders are attenuated
    assert abs(y[0, 0, 0, 0] - 0.5269763) < 1e-6 and abs(y[0, 0, 1, 0] - 0.7259313) < 1e-6


def test_hot_path_gaussian_constants():       # nsig (2,2)
    g = gaussian_kernel((3, 3), (2, 2))
    assert abs(g[1, 1] - 0.6193470306) < 1e-9 and abs(g[0, 1] - 0.0838195058) < 1e-9
    assert abs(g[0, 0] - 0.0113437366) < 1e-9 and abs(g.sum() - 1.0) < 1e-12


def test_ref_pyramid_shapes():                # tests/test_layer_blocks.py:118-131 (shapes of a 3-level split)
    cfg = OracleConfig((32, 32, 3), [4, 4, 4])
    x = np.random.default_rng(0).uniform(0, 255, (18, 32, 32, 3))
    bands = Oracle(cfg).pyramid(x)
    assert [tuple(b.shape) for b in bands] == [(18, 3, 32, 32), (18, 3, 16, 16), (18, 3, 8, 8)]


@pytest.mark.parametrize("H,W,k,s,ci,co", [(8, 8, 5, 2, 3, 4), (7, 9, 3, 2, 2, 3), (6, 6, 1, 1, 4, 4),
                                            (8, 8, 3, 1, 3, 5), (5, 5, 1, 2, 2, 2), (9, 9, 5, 2, 3, 2)])
def test_conv_restatements_agree(H, W, k, s, ci, co):
    rng = np.random.default_rng(0)
    x, b = rng.standard_normal((2, H, W, ci)), rng.standard_normal(co)
    w, wt = rng.standard_normal((k, k, ci, co)), rng.standard_normal((k, k, co, ci))
    xt = torch.tensor(x).permute(0, 3, 1, 2)
    y = conv2d_same(xt, torch.tensor(w), torch.tensor(b), (s, s)).permute(0, 2, 3, 1).numpy()
    assert np.abs(y - np_ops.conv2d_same_nhwc(x, w, b, (s, s))).max() < 1e-12
    y = conv2d_transpose_same(xt, torch.tensor(wt), torch.tensor(b), (s, s)).permute(0, 2, 3, 1).numpy()
    assert y.shape == (2, H * s, W * s, co)
    assert np.abs(y - np_ops.conv2d_transpose_same_nhwc(x, wt, b, (s, s))).max() < 1e-12


def test_conv_transpose_is_conv_adjoint():
    """Conv2DTranspose 'same' == VJP of the 'same' conv for input size n*s."""
    rng = np.random.default_rng(1)
    for k, s in [(5, 2), (3, 2), (1, 1), (3, 1), (1, 2)]:
        n = 6
        w = torch.tensor(rng.standard_normal((k, k, 3, 4)))
        xb = torch.tensor(rng.standard_normal((1, 3, n * s, n * s)), requires_grad=True)
        dy = torch.tensor(rng.standard_normal((1, 4, n, n)))
        (conv2d_same(xb, w, None, (s, s)) * dy).sum().backward()
        yt = conv2d_transpose_same(dy, w, None, (s, s))
        assert (yt - xb.grad).abs().max() < 1e-12


def test_bilinear_and_blur_restatements_agree():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 4, 6, 3))
    up = F.interpolate(torch.tensor(x).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear",
                       align_corners=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(np_ops.upsample2_bilinear_nhwc(x) - up).max() < 1e-12
    row = np_ops.upsample2_bilinear_nhwc(np.arange(4.0).reshape(1, 1, 4, 1))[0, 0, :, 0]
    assert np.allclose(row, [0, .25, .75, 1.25, 1.75, 2.25, 2.75, 3])      # SURVEY.md 8(c) recipe
    from oracle.mvae_oracle import gaussian_blur
    bl = gaussian_blur(torch.tensor(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    assert np.abs(np_ops.gaussian_blur_nhwc(x) - bl).max() < 1e-12


def test_depthwise_taps_equal_grouped_convolution():
    """The oracle writes DepthwiseConv2D (layer_blocks.py:604-614) and the fixed blur as nine shifted multiply-adds (speed:
    torch runs a float64 grouped convolution as C separate convolutions); same values and same gradients as
    F.conv2d(groups=C), and the same result as the loop-level numpy restatement."""
    from oracle.mvae_oracle import depthwise3x3_same
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 7, 9, dtype=torch.float64, generator=g, requires_grad=True)
    w = torch.randn(3, 3, 5, 1, dtype=torch.float64, generator=g, requires_grad=True)
    b = torch.randn(5, dtype=torch.float64, generator=g, requires_grad=True)
    y = depthwise3x3_same(x, w, b)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1)), w.permute(2, 3, 0, 1), b, groups=5)
    assert (y - ref).abs().max().item() <= 1e-14
    up = torch.randn(y.shape, dtype=torch.float64, generator=g)
    ga = torch.autograd.grad(y, (x, w, b), up, retain_graph=True)
    gb = torch.autograd.grad(ref, (x, w, b), up)
    for a, r in zip(ga, gb):
        assert (a - r).abs().max().item() <= 1e-13


def test_param_census_matches_survey():       # SURVEY.md 8 census table
    from tests.common import NB
    for dims, z, enc, want, ntens in [((32, 32, 3), [16] * 3, NB, 1295625, 420), ((32, 32, 3), [16] * 3, None, 2138313, None),
                                      ((256, 256, 3), [16] * 7, NB, 36045205, 980)]:
        P, _ = param_table(OracleConfig(dims, z, encoder=enc, decoder=enc))
        assert sum(int(np.prod(s)) for s, _ in P.values()) == want
        if ntens:
            assert len(P) == ntens


def test_oracle_gradients_match_finite_differences():
    """The oracle's autograd gradients are the ELBO's: central differences in float64 on the tiny model."""
    from tests.common import make_inputs, oracle_config, COMPILE
    oc = oracle_config("tiny")
    io = make_inputs("tiny", 3)
    orc = Oracle(oc)
    args = (io["state"], io["x"], io["eps"], io["noise"], io["keep"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    res, G = orc.loss_and_grads(io["params"], *args)
    rng = np.random.default_rng(5)
    for name in ["enc0.conv_base.w", "enc0.b0.mn.se.d1.w", "enc1.mu.w", "dec0.b0.convT.w", "dec1.bn.gamma", "dec0.out.b"]:
        p = {k: np.asarray(v, np.float64).copy() for k, v in io["params"].items()}
        idx = tuple(rng.integers(0, s) for s in p[name].shape)
        h = 1e-5
        p[name][idx] += h
        lp = orc.loss_and_grads(p, *args)[0]["loss"]
        p[name][idx] -= 2 * h
        lm = orc.loss_and_grads(p, *args)[0]["loss"]
        fd = (lp - lm) / (2 * h)
        assert abs(fd - G[name][idx]) <= 1e-4 * max(1.0, abs(fd)), (name, fd, G[name][idx])


def test_kink_mask_override_is_the_identity_on_the_oracles_own_active_sets():
    """Oracle.set_kink_masks (the subgradient choice a GPU parity test hands over): with the oracle's OWN active sets the
    gradients are unchanged and no unit is reported; flipping one ReLU unit is reported and changes the gradient."""
    from tests.common import COMPILE, make_inputs, oracle_config
    from oracle.mvae_oracle import Oracle
    name, B = "tiny", 4
    io = make_inputs(name, B)
    o = Oracle(oracle_config(name))
    inter = {}
    args = (io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"], COMPILE["r_loss_factor"],
            COMPILE["kl_loss_factor"])
    res, G = o.loss_and_grads(*args, inter=inter)
    masks = {}
    for k, v in inter.items():
        a = v.detach().numpy()
        if a.ndim == 4:
            a = np.transpose(a, (0, 2, 3, 1))
        if k.endswith((".t0", ".t1", ".s0")):
            masks[k] = (a > 0).reshape(B, -1)
        elif k.endswith(".ulin"):
            masks[k[:-5] + ".hsig"] = (a >= -2.5) & (a <= 2.5)
    assert any(k.endswith(".hsig") for k in masks) and any(k.endswith(".s0") for k in masks)
    o.set_kink_masks(masks)
    res2, G2 = o.loss_and_grads(*args)
    rep = o.kink_report()
    assert rep["flips"] == 0 and rep["units"] > 0
    assert res2["loss"] == res["loss"]
    for k in G:      # (to rounding: autograd adds the nine tap gradients of a depthwise layer in no fixed order)
        assert np.abs(G[k] - G2[k]).max() <= 1e-12 * max(1.0, np.abs(G[k]).max()), k
    key = next(k for k in masks if k.endswith(".t0"))
    flipped = dict(masks)
    flipped[key] = masks[key].copy()
    j = int(np.argmax(flipped[key].ravel()))            # an active unit
    flipped[key].reshape(-1)[j] = False
    o.set_kink_masks(flipped)
    res3, G3 = o.loss_and_grads(*args)
    assert o.kink_report()["flips"] == 1 and o.kink_report()["max_abs_at_flip"] > 0
    assert any(np.abs(G[k] - G3[k]).max() > 1e-9 * max(1.0, np.abs(G[k]).max()) for k in G)


def _block_params(kind, channels, filters=32, squeeze_units=-1, seed=0):
    from oracle.mvae_oracle import block_param_shapes
    rng = np.random.default_rng(seed)
    return {k: (0.3 * rng.standard_normal(shp)) for k, shp in block_param_shapes(kind, channels, filters, squeeze_units).items()}


def test_reference_shape_fixtures_of_the_hot_blocks():
    """tests/test_layer_blocks.py:76-113 of the reference: mobilenetV3_block(x, 32) and squeeze_excite_block(x, 32) on a
    (3, 256, 256, 3) input return (3, 256, 256, 3) -- filters / squeeze_units differ from the 3 input channels."""
    from oracle.mvae_oracle import mobilenetV3_block, squeeze_excite_block, block_param_shapes
    x = np.zeros((3, 256, 256, 3))
    p = _block_params("mnv3", 3, filters=32)
    assert p["conv0.w"].shape == (1, 1, 3, 32) and p["se.d0.w"].shape == (32, 32) and p["conv2.w"].shape == (1, 1, 32, 3)
    y = mobilenetV3_block(x, p)
    assert y.shape == (3, 256, 256, 3)
    ps = _block_params("se", 3, squeeze_units=32)
    assert ps["se.d0.w"].shape == (3, 32) and ps["se.d1.w"].shape == (32, 3)
    z = squeeze_excite_block(x, ps)
    assert z.shape == (3, 256, 256, 3)
    assert block_param_shapes("se", 3, squeeze_units=-1)["se.d0.w"] == (3, 3)           # layer_blocks.py:433-434
    with pytest.raises(ValueError):
        block_param_shapes("mnv3", 3, filters=0)                                          # layer_blocks.py:586-587


def test_general_block_equals_the_model_path_when_filters_equal_channels():
    """The stand-alone block with filters = channels, squeeze_units = -1 is what Oracle.mnv3 computes inside the model."""
    import torch
    from oracle.mvae_oracle import Oracle, mobilenetV3_block
    from tests.common import oracle_config
    c = 8
    p = _block_params("mnv3", c, filters=c, seed=3)
    x = np.random.default_rng(4).standard_normal((2, 6, 5, c))
    st = {"mean": 0.1 * np.arange(c), "var": 1.0 + 0.05 * np.arange(c)}
    y = mobilenetV3_block(x, p, training=False, state=st)
    o = Oracle(oracle_config("tiny"))
    T = {"b." + k: torch.as_tensor(v) for k, v in p.items()}
    S = {"b.se.bn.mean": torch.as_tensor(st["mean"]), "b.se.bn.var": torch.as_tensor(st["var"])}
    with torch.no_grad():
        ref = o.mnv3(torch.as_tensor(x).permute(0, 3, 1, 2), T, "b", S, False, None, {}, None).permute(0, 2, 3, 1).numpy()
    assert np.allclose(y, ref, rtol=1e-12, atol=1e-12)
