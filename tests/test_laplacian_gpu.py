"""GPU: the stand-alone Laplacian pyramid (mvae_laplacian_split / mvae_laplacian_merge through the facade) against the
CPU oracle, and the reference's round-trip property (tests/test_layer_blocks.py:161-190) at the bench batch size."""
import numpy as np
import pytest

from oracle import laplacian_oracle as lo

pytestmark = pytest.mark.gpu


def _models(dims, levels, **kw):
    from mvae import layer_blocks as lb
    split = lb.laplacian_transform_split(input_dims=dims, levels=levels, **kw)
    merge = lb.laplacian_transform_merge(input_dims=[(dims[0] >> i, dims[1] >> i, dims[2]) for i in range(levels)],
                                         levels=levels, **kw)
    return split, merge


@pytest.mark.parametrize("dims,levels,batch", [((32, 32, 3), 3, 18), ((16, 24, 1), 2, 5), ((64, 64, 3), 4, 3),
                                               ((8, 8, 2), 1, 4)])
def test_split_and_merge_match_the_oracle(dims, levels, batch):
    rng = np.random.default_rng(7)
    x = rng.uniform(0.0, 255.0, (batch,) + dims).astype(np.float32)
    split, merge = _models(dims, levels)
    got = split(x)
    want = lo.laplacian_split(x, levels)
    assert len(got) == levels
    for g, w in zip(got, want):
        assert g.shape == w.shape and g.dtype == np.float32
        assert np.abs(g - w).max() <= 2e-6                    # values are O(1): a few fp32 ulps
    bands = [rng.uniform(-1.0, 1.0, w.shape).astype(np.float32) for w in want]
    assert np.abs(merge(bands) - lo.laplacian_merge(bands)).max() <= 255.0 * 2e-6


def test_round_trip_at_bench_batch():
    x = np.random.default_rng(8).uniform(0.0, 255.0, (512, 32, 32, 3)).astype(np.float32)
    split, merge = _models((32, 32, 3), 3)
    back = merge(split(x))
    assert back.shape == x.shape
    assert np.all(np.abs(back - x)[:, 1:31, 1:31, :] <= 0.001)     # the reference's own bound
    assert np.abs(back - x).max() <= 1e-3


def test_value_range_and_errors():
    split, merge = _models((32, 32, 3), 3, min_value=-1.0, max_value=1.0)
    x = np.random.default_rng(9).uniform(-1.0, 1.0, (4, 32, 32, 3)).astype(np.float32)
    lv = split(x)
    assert np.abs(lv[2]).max() <= 1.0 + 1e-6                        # the coarsest level is a blurred, normalised image
    out = merge([10.0 * l for l in lv])                             # out-of-range sums are clipped (K.clip, :126-129)
    assert out.min() >= -1.0 and out.max() <= 1.0
    with pytest.raises(ValueError):
        split(np.zeros((2, 16, 16, 3), np.float32))
    from mvae import layer_blocks as lb
    with pytest.raises(ValueError):
        lb.laplacian_transform_split(input_dims=(30, 32, 3), levels=3)


@pytest.mark.parametrize("dims,levels,batch,filters", [((32, 32, 3), 3, 6, 32), ((16, 24, 1), 2, 3, 8)])
def test_trainable_merge_forward_matches_the_oracle(dims, levels, batch, filters):
    """laplacian_transform_merge(trainable=True) (layer_blocks.py:141-171): shapes as the reference's merge fixture
    (tests/test_layer_blocks.py:135-152), values against the oracle with the model's own and with perturbed weights."""
    from mvae import layer_blocks as lb
    shapes = [(dims[0] >> i, dims[1] >> i, dims[2]) for i in range(levels)]
    model = lb.laplacian_transform_merge(input_dims=shapes, levels=levels, trainable=True, filters=filters)
    rng = np.random.default_rng(5)
    bands = [rng.uniform(-1.0, 1.0, (batch,) + s).astype(np.float32) for s in shapes]
    w = model.get_weights()
    assert len(w) == levels - 1 and w[0]["mix.w"].shape == (3, 3, 2 * dims[2], filters) and w[0]["retarget.w"].shape == (1, 1, filters, dims[2])
    for scale in (1.0, 3.0):
        ws = [{k: (v * scale + (0.1 if k == "mix.b" else 0.0)).astype(np.float32) for k, v in d.items()} for d in w]
        model.set_weights(ws)
        got = model(bands)
        assert got.shape == (batch,) + dims and got.min() >= 0.0 and got.max() <= 255.0
        assert np.abs(got - lo.laplacian_merge_mix(bands, ws)).max() <= 255.0 * 5e-6
    with pytest.raises(ValueError):
        lb.laplacian_transform_merge(input_dims=shapes, levels=levels, trainable=True, activation="elu")
