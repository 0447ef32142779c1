"""Shared helpers for the test-suite: named configs, seeded inputs, oracle / HIP-engine runners."""
import os
import sys
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NB = {"filters": [64, 64, 64, 64, 32],
      "kernel_size": [(5, 5), (3, 3), (3, 3), (1, 1), (1, 1)],
      "strides": [(2, 2), (1, 1), (1, 1), (1, 1), (1, 1)]}          # notebooks/cifar10_notebook.ipynb:141-150

CONFIGS = {
    # SURVEY.md 8(c) "tiny" fixture model
    "tiny": dict(input_dims=(8, 8, 3), z_dims=[4, 4],
                 encoder={"filters": [8, 8], "kernel_size": [(3, 3), (1, 1)], "strides": [(2, 2), (1, 1)]},
                 decoder={"filters": [8, 8], "kernel_size": [(3, 3), (1, 1)], "strides": [(2, 2), (1, 1)]}),
    # odd shapes: rectangular image, 1 channel, non-square kernels / strides, channel counts that are not wave multiples
    "odd": dict(input_dims=(8, 16, 1), z_dims=[3, 5],
                encoder={"filters": [12, 20], "kernel_size": [(3, 5), (1, 1)], "strides": [(1, 2), (1, 1)]},
                decoder={"filters": [20, 12], "kernel_size": [(5, 3), (1, 1)], "strides": [(1, 2), (1, 1)]}),
    # constructor default encoder (multiscale_vae.py:17-21), decoder = reversed
    "c32def": dict(input_dims=(32, 32, 3), z_dims=[16, 16, 16], encoder=None, decoder=None),
    # main.py:81-91
    "main": dict(input_dims=(32, 32, 3), z_dims=[128, 64, 32],
                 encoder={"filters": [32, 32, 32], "kernel_size": [(3, 3)] * 3, "strides": [(2, 2), (2, 2), (1, 1)]},
                 decoder=None, sample_std=0.5),
    # BASELINE configs 1-3 (C32-nb)
    "c32nb": dict(input_dims=(32, 32, 3), z_dims=[16, 16, 16], encoder=NB, decoder=NB),
    # BASELINE configs 4-5 shrunk to 64x64 / 5 scales for parity runs (same block structure as C256-nb)
    "c64nb": dict(input_dims=(64, 64, 3), z_dims=[16] * 5, encoder=NB, decoder=NB),
    # the notebook blocks on a 24 x 40 image: widths 40 / 20 / 10 are not powers of two (the index-division paths of the
    # tiled conv kernels), 3 x 12 x 20 pixels do not fill the kernels' 128-pixel tiles, 40 columns need two ring strips
    "nbodd": dict(input_dims=(24, 40, 3), z_dims=[8, 8], encoder=NB, decoder=NB),
    # the notebook blocks on a 96 x 64 image: heights 96 / 48 / 24 are not powers of two, the 64-wide maps take two ring
    # strips (halo columns), the 32-wide ones one; rows per image 6144 / 1536 / 384 (the chained 1x1 launches need % 64)
    "c96nb": dict(input_dims=(96, 64, 3), z_dims=[16, 16], encoder=NB, decoder=NB),
    # BASELINE configs 4-5 (C256-nb), full size: 7 scales from 256x256 down to 4x4
    "c256nb": dict(input_dims=(256, 256, 3), z_dims=[16] * 7, encoder=NB, decoder=NB),
}
COMPILE = dict(learning_rate=0.001, r_loss_factor=1000.0, kl_loss_factor=10.0, clip_norm=1.0)   # notebook cells 5-6


def oracle_config(name):
    from oracle.mvae_oracle import OracleConfig
    c = dict(CONFIGS[name])
    return OracleConfig(c["input_dims"], c["z_dims"], encoder=c.get("encoder"), decoder=c.get("decoder"),
                        min_value=c.get("min_value", 0.0), max_value=c.get("max_value", 255.0),
                        sample_std=c.get("sample_std", 0.01))


def engine_args(name, max_batch):
    oc = oracle_config(name)
    return dict(input_dims=oc.input_dims, z_dims=oc.z_dims, encoder=oc.encoder, decoder=oc.decoder,
                min_value=oc.min_value, max_value=oc.max_value, sample_std=oc.sample_std, max_batch=max_batch)


def make_inputs(name, B, seed=0, perturb=True):
    """x ~ U[0,255); glorot params (+ perturbed biases / BN affine so no path is trivially zero);
    injected eps (scaled by sample_std), GaussianNoise draw, SpatialDropout keep mask."""
    from oracle.mvae_oracle import param_table
    from multiscale_variational_autoencoder_amd.initializers import init_params, init_state
    oc = oracle_config(name)
    P, S = param_table(oc)
    ptab = OrderedDict((k, dict(shape=v[0])) for k, v in P.items())
    stab = OrderedDict((k, dict(shape=v)) for k, v in S.items())
    rng = np.random.default_rng(1234 + seed)
    H, W, C = oc.input_dims
    x = rng.uniform(0.0, 255.0, (B, H, W, C)).astype(np.float32)
    params = init_params(ptab, seed=42 + seed)
    state = init_state(stab)
    if perturb:
        prng = np.random.default_rng(99 + seed)
        for k in params:
            if k.endswith(".b") or k.endswith(".beta"):
                params[k] = (0.1 * prng.standard_normal(params[k].shape)).astype(np.float32)
            elif k.endswith(".gamma"):
                params[k] = (1.0 + 0.1 * prng.standard_normal(params[k].shape)).astype(np.float32)
        for k in state:
            if k.endswith(".mean"):
                state[k] = (0.1 * prng.standard_normal(state[k].shape)).astype(np.float32)
            else:
                state[k] = (1.0 + 0.2 * prng.uniform(size=state[k].shape)).astype(np.float32)
    Z = sum(oc.z_dims)
    eps = (np.random.default_rng(7 + seed).standard_normal((B, Z)) * oc.sample_std).astype(np.float32)
    nrng = np.random.default_rng(11 + seed)
    noise = nrng.standard_normal((B, H, W, C)).astype(np.float32)
    keep = (nrng.uniform(size=(B, C)) >= 0.1).astype(np.float32)
    return dict(x=x, params=params, state=state, eps=eps, noise=noise, keep=keep)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm((a - b).ravel()) / (np.linalg.norm(b.ravel()) + 1e-30))


def reg_grad(params, ptab):
    """d(reg)/dw the optimiser kernel adds: 0.01*sign(w) ('l1'), 0.02*w ('l2')."""
    out = {}
    for k, meta in ptab.items():
        w = np.asarray(params[k], np.float64)
        reg = meta["reg"] if isinstance(meta, dict) else meta[1]
        out[k] = 0.01 * np.sign(w) if reg == "l1" else (0.02 * w if reg == "l2" else np.zeros_like(w))
    return out


def grad_errors(got, ref):
    """Per-tensor gradient error.  Criterion (written in the tests): ||got-ref|| <= max(TOL * ||ref||,
    2e-4 * rms * sqrt(n)) with rms the global per-element RMS of the reference gradient.  The second arm
    covers tensors whose true gradient is exactly zero (a bias feeding straight into BatchNorm): fp32
    summation leaves cancellation noise there, in the reference's own fp32 path as much as here.
    Returns name -> (err / max(||ref||, 0.1 * rms * sqrt(n))), so one threshold (2e-3) expresses both arms."""
    rms = np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in ref.values()) /
                  sum(np.asarray(g).size for g in ref.values()))
    out = {}
    for k, g in ref.items():
        g = np.asarray(g, np.float64)
        d = np.linalg.norm((np.asarray(got[k], np.float64) - g).ravel())
        out[k] = float(d / max(np.linalg.norm(g.ravel()), 0.1 * rms * np.sqrt(g.size)))
    return out


def structurally_zero(ref):
    """Names of tensors whose reference gradient vanishes identically (norm < 1e-6 of the global scale)."""
    rms = np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in ref.values()) /
                  sum(np.asarray(g).size for g in ref.values()))
    return {k for k, g in ref.items() if np.linalg.norm(np.asarray(g, np.float64).ravel()) < 1e-6 * rms * np.sqrt(np.asarray(g).size)}


def device_kink_masks(eng, B, x=None, min_value=0.0, max_value=255.0):
    """The active sets the device used at every kink of the graph (ReLU outputs > 0, hard_sigmoid inputs inside
    [-2.5, 2.5]), read back from the saved tensors of the last forward -- see Oracle.set_kink_masks.  With the input
    batch `x` also the branches of the loss: sign(y - recon) per pixel, the signs of the two channel-mean terms and the
    K.clip range mask (bfloat16 runs: the reconstruction error is large enough to flip those for a visible share of
    the pixels, float32 runs do not need them)."""
    masks = {}
    if x is not None:
        H, W, C = eng.input_dims
        recon = eng.tensor("recon", B).cpu().numpy().reshape(B, H, W, C)
        masks["loss.sign"] = np.sign(np.asarray(x, np.float32) - recon)
        sg = eng.tensor("loss_sgn", B).cpu().numpy().reshape(B, 2, C)
        masks["loss.ch_sign"], masks["loss.cc_sign"] = sg[:, 0], sg[:, 1]
        lin = (eng.tensor("merged0", B).cpu().numpy().reshape(B, H, W, C) + 1.0) * (max_value - min_value) * 0.5 + min_value
        masks["loss.clip"] = (lin >= min_value) & (lin <= max_value)
    for k in eng.param_table:
        if not k.endswith(".mn.conv0.w"):
            continue
        p = k[:-len(".conv0.w")]
        for t in ("t0", "t1", "s0"):
            masks[p + "." + t] = (eng.tensor(p + "." + t, B) > 0).cpu().numpy()
        u = eng.tensor(p + ".ulin", B)
        masks[p + ".hsig"] = ((u >= -2.5) & (u <= 2.5)).cpu().numpy()
    return masks


KINK_MAX_FRACTION = 2e-6     # at most this fraction of the units may sit on the other side of a kink than in the oracle
KINK_MAX_DISTANCE = 2e-5     # ... and only units this close to the kink (float32 forward error is ~1e-6 of O(1) values)


def check_kink_report(rep):
    assert rep["flips"] <= max(3, KINK_MAX_FRACTION * rep["units"]), rep
    assert rep["max_abs_at_flip"] <= KINK_MAX_DISTANCE, rep
