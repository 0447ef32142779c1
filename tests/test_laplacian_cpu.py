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
    with pytest.raises(NotImplementedError):
        lb.laplacian_transform_merge([(32, 32, 3), (16, 16, 3)], levels=2, trainable=True)
    import mvae
    assert mvae.layer_blocks.laplacian_transform_split is lb.laplacian_transform_split
