"""MVAE_DETERMINISTIC=1 (the stand-in for a race detector): every float reduction of the train step has a
fixed order, so two runs of the same step are BIT-identical -- gradients, losses, updated weights, BatchNorm state.  The
default mode (float atomics, split sums) stays within float32 summation noise of it."""
import os
import time

import numpy as np
import pytest

from tests.common import COMPILE, engine_args, make_inputs, rel_err

pytestmark = pytest.mark.gpu


def _steps(name, B, det, nsteps=2, production=False):
    from multiscale_variational_autoencoder_amd.engine import Engine
    os.environ["MVAE_DETERMINISTIC"] = "1" if det else "0"
    try:
        eng = Engine(**engine_args(name, B)).bind(0)
    finally:
        os.environ.pop("MVAE_DETERMINISTIC", None)
    assert eng.lib.mvae_deterministic(eng.h) == (1 if det else 0)
    io = make_inputs(name, B)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    grads = None
    t0 = time.time()
    for i in range(nsteps):
        if production:      # replayed hipGraphs, one stream per scale, on-device Philox
            eng.train_step(d["x"], COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                           COMPILE["clip_norm"], seed=100 + i)
        else:
            out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "losses"))
            eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
            if i == 0:
                grads = eng.get_grads()
                first = {k: v.cpu().numpy().copy() for k, v in out.items()}
            eng.apply(COMPILE["learning_rate"], COMPILE["clip_norm"])
    eng.sync()
    dt = time.time() - t0
    res = dict(params=eng.get_params(), state=eng.get_state(), accum=eng.get_accum(), grads=grads, dt=dt)
    if not production:
        res.update(first)
    eng.close()
    return res


@pytest.mark.parametrize("name,B", [("c32nb", 16), ("c64nb", 4)])
def test_two_deterministic_runs_are_bit_identical(name, B):
    a = _steps(name, B, True)
    b = _steps(name, B, True)
    for group in ("grads", "params", "state", "accum"):
        for k in a[group]:
            assert np.array_equal(a[group][k], b[group][k]), (group, k, float(np.abs(a[group][k] - b[group][k]).max()))
    assert np.array_equal(a["recon"], b["recon"]) and np.array_equal(a["losses"], b["losses"])
    # ... and the default mode computes the same numbers up to float32 summation order
    c = _steps(name, B, False)
    gmax = max(np.linalg.norm(v) for v in a["grads"].values())
    # One ReLU unit whose pre-activation is within rounding of zero can land on the other side in the default run (its
    # forward sums are float atomics; DESIGN.md section 2, kinks): at batch 4 that moves the dozen gradient tensors behind it
    # by 1e-3 .. 2e-3 -- in any build, the float32-MFMA kernels included (tools/det_spread.py: two DEFAULT runs differ by the
    # same amounts).  So: nearly every tensor within 2e-4, none beyond the kink level.
    errs = {k: rel_err(c["grads"][k], a["grads"][k]) for k in a["grads"] if np.linalg.norm(a["grads"][k]) > 1e-3 * gmax}
    vals = np.array(list(errs.values()))
    worst = max(errs, key=errs.get)
    assert np.percentile(vals, 90) <= 2e-4 and (vals > 2e-4).sum() <= 24 and vals.max() <= 5e-3, (worst, errs[worst], int((vals > 2e-4).sum()))
    assert np.abs(c["recon"] - a["recon"]).max() <= 1e-4 * 255
    assert rel_err(c["losses"], a["losses"]) <= 1e-5


def test_deterministic_production_path_and_cost():
    """The replayed-graph, multi-stream, device-RNG path is deterministic too; the mode's step-time cost is printed."""
    a = _steps("c32nb", 64, True, nsteps=6, production=True)
    b = _steps("c32nb", 64, True, nsteps=6, production=True)
    for k in a["params"]:
        assert np.array_equal(a["params"][k], b["params"][k]), k
    c = _steps("c32nb", 64, False, nsteps=6, production=True)
    print("deterministic mode: %.1f ms/step against %.1f ms/step (6 steps of C32-nb at batch 64, incl. graph capture)" %
          (1e3 * a["dt"] / 6, 1e3 * c["dt"] / 6))
