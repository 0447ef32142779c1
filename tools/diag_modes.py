"""Diagnostic: compare execution modes (graphs / streams on-off) on identical seeds; report the tensors that differ."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import CONFIGS, engine_args
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params

name, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
x = np.random.default_rng(5).uniform(0, 255, (B,) + tuple(CONFIGS[name]["input_dims"])).astype(np.float32)
runs = {}
for mode in (("0", "0"), ("0", "0"), ("0", "1"), ("1", "0"), ("1", "1")):
    os.environ["MVAE_GRAPHS"], os.environ["MVAE_STREAMS"] = mode
    eng = Engine(**engine_args(name, B)).bind()
    p0 = init_params(eng.param_table, 42)
    eng.set_params(p0)
    xd = eng.to_device(x)
    for s in range(steps):
        eng.train_step(xd, 1e-3, 1000.0, 10.0, 1.0, seed=100 + s)
    key = "g%s_s%s" % mode + ("_b" if ("g%s_s%s" % mode) in runs else "")
    runs[key] = (eng.get_params(), eng.get_grads())
ref = runs["g0_s0"][0]
flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in ref])
trav = np.linalg.norm(flat(ref) - flat(p0))
for k, (p, g) in runs.items():
    d = np.linalg.norm(flat(p) - flat(ref))
    per = sorted(((float(np.linalg.norm(p[t].astype(np.float64) - ref[t])), t) for t in ref), reverse=True)[:4]
    print("%-10s dist/travelled %.3e   worst tensors: %s" % (k, d / trav, [(t, "%.2e" % v) for v, t in per]))
