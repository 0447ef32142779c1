"""CPU: host logic of the drop-in facade -- constructor checks and messages of mvae/multiscale_vae.py:34-45 and
layer_blocks.py:918-927, decoder default, LR schedule (mvae/schedule.py:17-19), glorot initialisation, model views,
weight save/load, and that the HIP path is the only compute path."""
import json
import os

import numpy as np
import pytest

from tests.common import NB


def _vae(**kw):
    from multiscale_variational_autoencoder_amd import MultiscaleVAE
    args = dict(input_dims=(32, 32, 3), z_dims=[16, 16, 16])
    args.update(kw)
    return MultiscaleVAE(**args)


def test_constructor_contract(hip_lib):
    import inspect
    from multiscale_variational_autoencoder_amd import MultiscaleVAE
    sig = inspect.signature(MultiscaleVAE.__init__)
    names = [p for p in sig.parameters][1:10]
    assert names == ["input_dims", "z_dims", "compress_output", "encoder", "decoder", "min_value", "max_value",
                     "sample_std", "channels_index"]                       # multiscale_vae.py:12-26, verbatim order
    assert sig.parameters["sample_std"].default == 0.01 and sig.parameters["max_value"].default == 255.0
    with pytest.raises(ValueError, match="encoder cannot be None"):
        _vae(encoder=None)
    with pytest.raises(ValueError, match="z_dims elements should be > 0"):
        _vae(z_dims=[16, -1, 4])
    with pytest.raises(ValueError, match=r"len\(filters\) \[2\] should be equal to len\(kernel_size\) \[1\]"):
        _vae(encoder={"filters": [32, 32], "kernel_size": [(3, 3)], "strides": [(1, 1), (1, 1)]})
    v = _vae(encoder={"filters": [32, 64], "kernel_size": [(3, 3), (5, 5)], "strides": [(2, 2), (1, 1)]})
    assert v._decoder_config == {"filters": [64, 32], "kernel_size": [(5, 5), (3, 3)], "strides": [(1, 1), (2, 2)]}
    import mvae                                                             # `from mvae import MultiscaleVAE` still works
    assert mvae.MultiscaleVAE is MultiscaleVAE


def test_surface_of_the_reference_and_its_callers(hip_lib):
    v = _vae(encoder=NB, decoder=NB)
    for attr in ("compile", "train", "fit", "predict", "load_weights", "normalize", "encoder", "decoder",
                 "model_trainable", "model_encode", "model_decode", "model_predict", "learning_rate"):
        assert hasattr(v, attr), attr
    v.compile(learning_rate=0.001, r_loss_factor=1000, kl_loss_factor=10)
    assert v.learning_rate == 0.001
    v.learning_rate = 0.5
    assert v.learning_rate == 0.5
    assert np.allclose(v.normalize(np.array([0.0, 127.5, 255.0])), [0.0, 0.5, 1.0])       # :586-587
    for m in (v.model_trainable, v.encoder, v.decoder):
        assert callable(m.predict) and callable(m.summary) and callable(m.to_json)
        assert json.loads(m.to_json())["config"]["z_dims"] == [16, 16, 16]
    lines = []
    v.model_trainable.summary(print_fn=lines.append)
    assert any("1,295,625" in l for l in lines)                            # SURVEY.md census, C32-nb
    assert v.encoder.count_params() + v.decoder.count_params() == 1295625
    with pytest.raises(RuntimeError):
        v.encoder.fit(np.zeros((2, 32, 32, 3)))


def test_no_cpu_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multiscale_variational_autoencoder_amd import MvaeError
    v = _vae()
    with pytest.raises(MvaeError, match="no CPU fallback"):
        v.predict(np.zeros((2, 32, 32, 3), np.float32))
    v.compile(1e-3)
    with pytest.raises(MvaeError, match="no CPU fallback"):
        v.train(np.zeros((4, 32, 32, 3), np.float32), batch_size=2, epochs=1, run_folder=None)


def test_step_decay_schedule():
    from multiscale_variational_autoencoder_amd.schedule import step_decay, step_decay_schedule
    from oracle.mvae_oracle import step_decay as ref
    f = step_decay(0.001, 0.5, 10)                                          # notebook cell 9
    for e in (0, 1, 9, 10, 19, 20, 49):
        assert f(e) == pytest.approx(ref(0.001, 0.5, 10, e)) == pytest.approx(0.001 * 0.5 ** (e // 10))
    class V:
        learning_rate = None
    cb = step_decay_schedule(0.01, 0.75, 20)                                # main.py:119-127
    v = V(); cb.set_vae(v); cb.on_epoch_begin(45)
    assert v.learning_rate == pytest.approx(0.01 * 0.75 ** 2)


def test_glorot_normal_initialisation(hip_lib):
    v = _vae(encoder=NB, decoder=NB)
    w = v._weights
    k = w["dec0.b1.mn.conv0.w"]                                             # (1,1,64,64): fan_in = fan_out = 64
    assert abs(k.std() / np.sqrt(2.0 / 128) - 1) < 0.05 and np.abs(k).max() <= 2 * np.sqrt(2.0 / 128) / 0.8796 + 1e-6
    d = w["enc0.b0.mn.dw.w"]                                                # depthwise (3,3,64,1): fans 576 / 9
    assert abs(d.std() / np.sqrt(2.0 / (576 + 9)) - 1) < 0.1
    m = w["enc0.mu.w"]                                                      # Dense (8192,16)
    assert abs(m.std() / np.sqrt(2.0 / (8192 + 16)) - 1) < 0.02
    assert all(np.all(w[k] == 0) for k in w if k.endswith(".b") or k.endswith(".beta"))
    assert all(np.all(w[k] == 1) for k in w if k.endswith(".gamma"))
    assert all(np.all(s == (1 if k.endswith(".var") else 0)) for k, s in v._state.items())
    v2 = _vae(encoder=NB, decoder=NB)
    assert all(np.array_equal(w[k], v2._weights[k]) for k in w)             # seeded: identical replicas


def test_weights_roundtrip_on_host(hip_lib, tmp_path):
    v = _vae()
    path = str(tmp_path / "w.npz")
    v.save_weights(path)
    v2 = _vae(seed=7)
    assert not np.array_equal(v2._weights["enc0.mu.w"], v._weights["enc0.mu.w"])
    v2.load_weights(path)
    assert all(np.array_equal(v2._weights[k], v._weights[k]) for k in v._weights)
    v2.load_weights(str(tmp_path / "missing.h5"))                           # reference load_weights is a no-op stub


def test_collage_resize_and_png_writer(tmp_path):
    """callbacks.py helpers (the reference's callback needs a `collage` it never defines, mvae/callbacks.py:10,56)."""
    import struct
    import zlib
    from multiscale_variational_autoencoder_amd.callbacks import collage, resize_nearest, save_png
    x = np.arange(5 * 2 * 3 * 3, dtype=np.float32).reshape(5, 2, 3, 3)
    c = collage(x)                                         # 5 images -> 3 x 2 grid
    assert c.shape == (2 * 2, 3 * 3, 3)
    assert np.array_equal(c[0:2, 3:6], x[1]) and np.array_equal(c[2:4, 0:3], x[3]) and np.all(c[2:4, 6:9] == 0)
    assert collage(x[..., :1]).shape == (4, 9)             # single channel -> 2-D
    r = resize_nearest(np.array([[0.0, 1.0], [2.0, 3.0]]), (4, 6))
    assert r.shape == (4, 6) and np.array_equal(r[0], [0, 0, 0, 1, 1, 1]) and np.array_equal(r[:, 0], [0, 0, 2, 2])
    path = str(tmp_path / "t.png")
    img = np.linspace(0, 1, 4 * 5 * 3).reshape(4, 5, 3)
    save_png(path, img)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    w, h, depth, ctype = struct.unpack(">IIBB", raw[16:26])
    assert (w, h, depth, ctype) == (5, 4, 8, 2)
    n = struct.unpack(">I", raw[33:37])[0]
    assert raw[37:41] == b"IDAT"
    rows = zlib.decompress(raw[41:41 + n])
    assert len(rows) == 4 * (1 + 5 * 3) and rows[0] == 0
    assert np.array_equal(np.frombuffer(rows, np.uint8).reshape(4, 16)[:, 1:].reshape(4, 5, 3), np.rint(img * 255).astype(np.uint8))
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(path)), np.rint(img * 255).astype(np.uint8))
    except ImportError:
        pass


def test_shard_batches_drops_nothing_and_keeps_ranks_in_step():
    """Data-parallel feed: every sample of every global batch is taken by some rank, all ranks take the same
    number of steps with the same local batch sizes, the tail wraps instead of being dropped."""
    from multiscale_variational_autoencoder_amd.multiscale_vae import shard_batches
    n, bs = 103, 16                      # 6 full batches + a tail of 7
    order = np.random.default_rng(0).permutation(n)
    for world in (1, 2, 3, 8):
        shards = [shard_batches(order, bs, world, r) for r in range(world)]
        spans = [s[1] for s in shards]
        assert all(sp == spans[0] for sp in spans)                       # same steps, same local batch sizes
        assert len(spans[0]) == -(-n // bs)
        for b, (off, cnt) in enumerate(spans[0]):
            g = order[b * bs:(b + 1) * bs]
            taken = np.concatenate([sh[0][off:off + cnt] for sh in shards])
            assert set(taken.tolist()) == set(g.tolist())                # nothing dropped
            assert len(taken) == world * (-(-len(g) // world))            # padded by wrap-around only
            assert len(taken) - len(g) < world
    local, spans = shard_batches(order, bs, 1, 0)
    assert np.array_equal(local, order) and sum(c for _, c in spans) == n
