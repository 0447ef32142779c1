"""CPU: the Laplacian-pyramid oracle against the reference's own tests for laplacian_transform_split / _merge
(tests/test_layer_blocks.py:118-190), and the host-side argument checks of the facade (no GPU needed)."""
import numpy as np
import pytest

from oracle import laplacian_oracle as lo


def test_split_shapes_like_the_reference_test():
    x = np.random.default_rng(0).uniform(0.0, 255.0, (18, 32, 32, 3))
    res = lo.laplacian_split(x, 3)                                  # test_layer_blocks.py:118-131
    assert [r.shape for r in res] == [(18, 32, 32, 3), (18, 16, 16, 3), (18, 8, 8, 3)]


def test_merge_shape_like_the_reference_test():
    rng = np.random.default_rng(1)
    xs = [rng.uniform(0.0, 255.0, (18, 32 >> i, 32 >> i, 3)) for i in range(3)]
    assert lo.laplacian_merge(xs).shape == (18, 32, 32, 3)          # test_layer_blocks.py:137-155


def test_split_merge_round_trip_like_the_reference_test():
    x = np.random.default_rng(2).uniform(0.0, 255.0, (18, 32, 32, 3))
    back = lo.laplacian_merge(lo.laplacian_split(x, 3))             # test_layer_blocks.py:161-190
    assert np.all(np.abs(back - x)[:, 1:31, 1:31, :] <= 0.001)
    assert np.abs(back - x).max() <= 1e-9                            # the restatement is invertible everywhere


def test_upsample_is_half_pixel_bilinear():
    x = np.arange(4, dtype=np.float64).reshape(1, 2, 2, 1)
    up = lo.upsample2_bilinear(x)[0, :, :, 0]
    assert np.allclose(up[0], [0.0, 0.25, 0.75, 1.0])                # rows clamp at the edge, columns 0.75/0.25 blends
    assert np.allclose(up[:, 0], [0.0, 0.5, 1.5, 2.0])


def test_facade_argument_checks():
    from multiscale_variational_autoencoder_amd import layer_blocks as lb
    assert np.allclose(lb.gaussian_kernel((3, 3), (1, 1)), lo.gaussian_kernel((3, 3), (1, 1)))
    m = lb.laplacian_transform_merge([(32, 32, 3), (16, 16, 3)], levels=2, trainable=True)      # host-side build only
    w = m.get_weights()
    assert len(w) == 1 and w[0]["mix.w"].shape == (3, 3, 6, 32) and w[0]["retarget.w"].shape == (1, 1, 32, 3)
    with pytest.raises(ValueError):
        m.set_weights([{"mix.w": np.zeros((3, 3, 6, 8)), "mix.b": np.zeros(8), "retarget.w": np.zeros((1, 1, 8, 3))}])
    with pytest.raises(ValueError):
        lb.laplacian_transform_merge([(32, 32, 3), (16, 16, 3)], levels=2, trainable=True, filters=0)
    import mvae
    assert mvae.layer_blocks.laplacian_transform_split is lb.laplacian_transform_split


def test_trainable_merge_oracle_reduces_to_the_plain_merge_with_zero_mixing_weights():
    """tanh(0) = 0: with the retargeting conv zeroed, merge(trainable=True) is ... + level only, i.e. the levels' own sum
    without the upsampled carry -- and with levels = 1 both merges are the denormalised input (layer_blocks.py:137-171)."""
    import numpy as np
    from oracle import laplacian_oracle as lo
    rng = np.random.default_rng(0)
    bands = [rng.uniform(-1, 1, (2, 8, 8, 3)), rng.uniform(-1, 1, (2, 4, 4, 3))]
    w = [{"mix.w": rng.standard_normal((3, 3, 6, 5)), "mix.b": np.zeros(5), "retarget.w": np.zeros((1, 1, 5, 3))}]
    assert np.allclose(lo.laplacian_merge_mix(bands, w), np.clip((bands[0] + 1.0) * 127.5, 0, 255))
    assert np.allclose(lo.laplacian_merge_mix(bands[:1], []), lo.laplacian_merge(bands[:1]))
    # SAME-conv restatement against the depthwise Gaussian of the same module (a 3x3 conv with a diagonal kernel)
    g = lo.gaussian_kernel((3, 3), (1, 1))
    wk = np.zeros((3, 3, 3, 3))
    for c in range(3):
        wk[:, :, c, c] = g
    x = rng.uniform(-1, 1, (2, 8, 8, 3))
    assert np.allclose(lo._conv_same(x, wk), lo.gaussian_filter(x))
