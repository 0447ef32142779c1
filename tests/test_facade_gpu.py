"""GPU: the drop-in surface end to end -- what a user of the reference's notebook does (cifar10_notebook.ipynb cells
4-10): construct, compile, train with the step-decay schedule and checkpoints, predict / encode / decode, then
resume from a checkpoint.  Everything here runs through `MultiscaleVAE`, i.e. ctypes -> libmvae_hip.so; the oracle is
used once, to check the facade's train_on_batch against one oracle train step."""
import glob
import os

import numpy as np
import pytest

from tests.common import COMPILE, CONFIGS, make_inputs, oracle_config, rel_err, structurally_zero

pytestmark = pytest.mark.gpu

SMALL = dict(input_dims=(16, 16, 3), z_dims=[8, 8],
             encoder={"filters": [32, 32], "kernel_size": [(3, 3), (3, 3)], "strides": [(2, 2), (1, 1)]})


def _vae(**kw):
    from multiscale_variational_autoencoder_amd import MultiscaleVAE
    args = dict(SMALL)
    args.update(kw)
    return MultiscaleVAE(**args)


def _data(n, seed=3):
    rng = np.random.default_rng(seed)
    base = rng.uniform(0, 255, (n, 1, 1, 3))
    return np.clip(base + rng.normal(0, 20, (n, 16, 16, 3)), 0, 255).astype(np.float32)


def test_train_predict_encode_decode(tmp_path):
    v = _vae()
    v.compile(learning_rate=0.01, r_loss_factor=1000, kl_loss_factor=10)
    x = _data(96)
    hist = v.train(x, batch_size=32, epochs=4, run_folder=str(tmp_path), step_size=2, lr_decay=0.5,
                   save_checkpoint_weights=True)
    assert hist.epoch == [0, 1, 2, 3]
    for k in ("loss", "vae_r_loss", "vae_kl_loss", "images_per_sec"):
        assert len(hist.history[k]) == 4 and np.all(np.isfinite(hist.history[k])), k
    assert hist.history["loss"][-1] < hist.history["loss"][0]              # it learns
    assert v.learning_rate == pytest.approx(0.01 * 0.5)                      # schedule.py:17-19 at epoch 3: floor(3/2)=1
    assert len(glob.glob(os.path.join(str(tmp_path), "weights", "weights-*.npz"))) == 4
    # the intermediate-results callback (mvae/callbacks.py:66-135): every 100 batches -> batch 0 of each epoch here
    for prefix in ("img", "samples", "interpolations"):
        files = sorted(glob.glob(os.path.join(str(tmp_path), "images", prefix + "_*.png")))
        assert [os.path.basename(f) for f in files] == ["%s_%03d_0.png" % (prefix, e) for e in (1, 2, 3, 4)], files
        assert open(files[0], "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    recon = v.model_trainable.predict(x[:10], batch_size=4)                  # last partial batch of 2
    assert recon.shape == (10, 16, 16, 3) and recon.dtype == np.float32
    assert recon.min() >= 0.0 and recon.max() <= 255.0                       # denormalize clips (multiscale_vae.py:86-94)
    z = v.encoder.predict(x[:10])
    assert z.shape == (10, 16)
    dec = v.decoder.predict(z)
    assert dec.shape == (10, 16, 16, 3)
    # decode(encode(x)) is the trainable model's inference path up to the sampling noise (sample_std = 0.01)
    assert np.abs(dec - recon).mean() < 0.05 * 255
    assert np.allclose(v.model_predict.predict(x[:0]).shape, (0, 16, 16, 3))


def test_checkpoint_resume_reproduces_the_run(tmp_path):
    x = _data(64, seed=5)
    a = _vae()
    a.compile(learning_rate=0.01, r_loss_factor=1000, kl_loss_factor=10)
    a.train(x, batch_size=16, epochs=1, run_folder=None)
    ck = str(tmp_path / "epoch1.npz")
    a.save_weights(ck)
    a.train(x, batch_size=16, epochs=2, run_folder=None, initial_epoch=1)
    wa = a.get_weights()

    b = _vae()
    b.compile(learning_rate=0.01, r_loss_factor=1000, kl_loss_factor=10)
    b.load_weights(ck)                                                       # weights, Adagrad accumulators, BN state, RNG step
    for k, v in np.load(ck).items():
        if k.startswith("w/"):
            assert np.array_equal(b.get_weights()[k[2:]], v)
    b.train(x, batch_size=16, epochs=2, run_folder=None, initial_epoch=1)
    wb = b.get_weights()
    # same shuffles, same device RNG stream, same optimiser state: the resumed epoch retraces the original one up to
    # float-atomic summation order.  Measured against the distance the parameters travel in that epoch (tensors whose
    # true gradient is identically zero -- biases feeding BatchNorm -- random-walk on rounding noise, so a per-tensor
    # relative error is meaningless for them; an optimiser state that was NOT restored moves the result by O(1) here).
    w0 = {k[2:]: v for k, v in np.load(ck).items() if k.startswith("w/")}
    flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in wa])
    travelled = np.linalg.norm(flat(wa) - flat(w0))
    dist = np.linalg.norm(flat(wb) - flat(wa))
    assert travelled > 0 and dist <= 0.15 * travelled, (dist, travelled)
    # a run resumed WITHOUT the Adagrad accumulators / RNG position does not retrace it
    c = _vae()
    c.compile(learning_rate=0.01, r_loss_factor=1000, kl_loss_factor=10)
    c.set_weights(w0)
    c.train(x, batch_size=16, epochs=2, run_folder=None, initial_epoch=1)
    assert np.linalg.norm(flat(c.get_weights()) - flat(wa)) > 2.0 * dist


def test_train_on_batch_matches_one_oracle_step():
    from oracle.mvae_oracle import Oracle
    from multiscale_variational_autoencoder_amd import MultiscaleVAE
    name, B = "tiny", 8
    cfg = CONFIGS[name]
    io = make_inputs(name, B)
    v = MultiscaleVAE(input_dims=cfg["input_dims"], z_dims=cfg["z_dims"], encoder=cfg["encoder"],
                      decoder=cfg.get("decoder"), min_value=cfg.get("min_value", 0.0), max_value=cfg.get("max_value", 255.0),
                      sample_std=cfg.get("sample_std", 0.01))
    v.set_weights(io["params"])
    v.compile(COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], COMPILE["clip_norm"])
    eng = v.train_on_batch(io["x"], eps=io["eps"], noise=io["noise"], keep_mask=io["keep"])
    eng.sync()
    o = Oracle(oracle_config(name))
    p0 = {k: np.asarray(val, np.float64) for k, val in io["params"].items()}
    a0 = {k: np.full(val.shape, 0.1) for k, val in p0.items()}                 # Adagrad initial accumulator
    st = {k: np.asarray(val, np.float64) for k, val in io["state"].items()}
    res, G, p, acc, _ = o.train_step(p0, a0, st, io["x"], io["eps"], io["noise"], io["keep"], COMPILE["learning_rate"],
                                     COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], COMPILE["clip_norm"])
    w = v.get_weights()
    zero = structurally_zero(G)          # noise-only gradients: Adagrad turns them into noise-sized steps
    # error measured against the size of the step (lr), as in test_adagrad_trajectory_parity
    worst = max(float(np.abs(w[k] - p[k]).max() / COMPILE["learning_rate"]) * (0.1 if k in zero else 1.0) for k in p)
    assert worst <= 3e-2, worst


def test_bf16_facade_train_checkpoint_predict(tmp_path):
    """The same notebook flow with act_dtype="bf16" on a 32x32 model whose scales run in bfloat16: trains (the loss
    comes down), predicts in range, saves a checkpoint a float32 model of the same architecture loads (parameters are
    float32 in both modes), and the two models then agree on the reconstruction within the bf16 forward bar."""
    from multiscale_variational_autoencoder_amd import MultiscaleVAE
    cfg = CONFIGS["c32nb"]
    args = dict(input_dims=cfg["input_dims"], z_dims=cfg["z_dims"], encoder=cfg["encoder"], decoder=cfg["decoder"])
    v = MultiscaleVAE(act_dtype="bf16", **args)
    assert v._engine.scale_dtypes()[:2] == ["bf16", "bf16"]
    v.compile(learning_rate=0.003, r_loss_factor=1000, kl_loss_factor=10)
    rng = np.random.default_rng(4)
    base = rng.uniform(0, 255, (256, 1, 1, 3))
    x = np.clip(base + rng.normal(0, 20, (256, 32, 32, 3)), 0, 255).astype(np.float32)
    hist = v.train(x, batch_size=64, epochs=3, run_folder=str(tmp_path), step_size=2, lr_decay=0.5,
                   save_checkpoint_weights=True)
    assert np.all(np.isfinite(hist.history["loss"])) and hist.history["loss"][-1] < hist.history["loss"][0]
    recon = v.model_trainable.predict(x[:16], batch_size=16)
    assert recon.shape == (16, 32, 32, 3) and recon.min() >= 0.0 and recon.max() <= 255.0
    ck = sorted(glob.glob(os.path.join(str(tmp_path), "weights", "weights-*.npz")))[-1]
    w = MultiscaleVAE(**args)                                # float32 activations, same parameters
    w.compile(learning_rate=0.003, r_loss_factor=1000, kl_loss_factor=10)
    w.load_weights(ck)
    recon32 = w.model_trainable.predict(x[:16], batch_size=16)
    d = recon.astype(np.float64) - recon32
    assert np.sqrt((d ** 2).mean()) <= 1.5e-2 * 255 and np.abs(d).max() <= 0.15 * 255, (np.sqrt((d ** 2).mean()), np.abs(d).max())
