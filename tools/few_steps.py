#!/usr/bin/env python3
"""A handful of train steps and nothing else (for AMD_LOG_LEVEL runs):  python tools/few_steps.py [workload] [batch] [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params
name = sys.argv[1] if len(sys.argv) > 1 else "c32nb"
w = WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else w["batch"]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
eng = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype=w.get("dtype", "f32")).bind(0)
eng.set_params(init_params(eng.param_table, 42))
x = eng.to_device(np.random.default_rng(0).uniform(0, 255, (B,) + tuple(w["input_dims"])).astype(np.float32))
for step in range(n):
    sys.stderr.write("=== step %d\n" % step); sys.stderr.flush()
    eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=step)
eng.sync()
