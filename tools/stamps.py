#!/usr/bin/env python3
"""Unperturbed timeline of a train step from in-graph device time stamps (MVAE_STAMPS=1; csrc/runtime.cpp stamp()):
when each scale's forward / backward chain starts and ends relative to the step start.  GPU box:
    MVAE_STAMPS=1 python tools/stamps.py [c32nb|c256nb] [batch] [f32|bf16]
STAMPS_FREE_RUNNING=1: the loop never synchronises (as a training loop) and the stamps of its last step are shown -- reading the
stamps after every step starts each step on an idle device, with the graph launch's node-by-node enqueue visible as late chain starts."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MVAE_STAMPS", "1")
from bench import WORKLOADS                                                     # noqa: E402
from multiscale_variational_autoencoder_amd.engine import Engine                # noqa: E402
from multiscale_variational_autoencoder_amd.initializers import init_params     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c32nb"
w = WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else w["batch"]
act = sys.argv[3] if len(sys.argv) > 3 else w.get("dtype", "f32")
eng = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype=act).bind(0)
eng.set_params(init_params(eng.param_table, 42))
x = eng.to_device(np.random.default_rng(0).uniform(0, 255, (B,) + tuple(w["input_dims"])).astype(np.float32))
L = len(w["z_dims"])
runs = []
FREE = os.environ.get("STAMPS_FREE_RUNNING", "") == "1"     # no synchronisation between steps: the stamps of the LAST of 60 steps
if FREE:
    x = eng.stage_input(x)
    for step in range(60):
        eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=step)
    eng.sync()
    buf = (C.c_uint64 * 64)()
    assert eng.lib.mvae_stamps(eng.h, buf, 64) == 0
    runs.append(np.array(buf[:], dtype=np.float64))
for step in range(0 if FREE else 30):
    eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=step)
    if step >= 20:
        buf = (C.c_uint64 * 64)()
        rc = eng.lib.mvae_stamps(eng.h, buf, 64)
        assert rc == 0, rc
        runs.append(np.array(buf[:], dtype=np.float64))
t = np.median(np.stack(runs), axis=0)
us = lambda i: (t[i] - t[0]) / 100.0                                            # 100 MHz clock
print("%s B=%d %s   (median of %d steps; microseconds after the forward's first kernel)" % (name, B, act, len(runs)))
print("forward : fork %.0f  joined %.0f  end %.0f" % (us(1), us(2), us(3)))
for i in range(L):
    print("   scale %d forward  %7.0f .. %7.0f  (%6.0f us)" % (i, us(10 + i), us(20 + i), us(20 + i) - us(10 + i)))
print("backward: start %.0f  fork %.0f  joined %.0f  end %.0f" % (us(4), us(5), us(6), us(7)))
for i in range(L):
    print("   scale %d backward %7.0f .. %7.0f  (%6.0f us)" % (i, us(30 + i), us(40 + i), us(40 + i) - us(30 + i)))
print("apply   : %.0f .. %.0f" % (us(8), us(9)))
