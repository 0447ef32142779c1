#!/usr/bin/env python3
"""Host time of the three ABI calls of a train step (how long the CPU thread spends inside mvae_forward / mvae_backward /
mvae_apply_adagrad = hipGraphLaunch of their captured graphs) next to the device time of a step.  CAREFUL: a loop that
never synchronises runs ahead of the device until the AQL ring is full, after which every launch waits for the device --
the host then appears to need one device step per step whatever the launch costs.  Only "loop without sync" < "with final
sync" (the host got ahead) says the step is NOT launch-bound; the launch cost itself is in tools/graph_launch_cost.hip.
    python tools/host_times.py [c32nb] [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params
name = sys.argv[1] if len(sys.argv) > 1 else "c32nb"
w = WORKLOADS[name]
B = int(sys.argv[2]) if len(sys.argv) > 2 else w["batch"]
act = sys.argv[3] if len(sys.argv) > 3 else w.get("dtype", "f32")
eng = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype=act).bind(0)
eng.set_params(init_params(eng.param_table, 42))
x = eng.to_device(np.random.default_rng(0).uniform(0, 255, (B,) + tuple(w["input_dims"])).astype(np.float32))
torch = eng.torch
for step in range(30):
    eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=step)
torch.cuda.synchronize()
N = 200
tf = tb = ta = 0.0
t_all0 = time.perf_counter()
for step in range(N):
    t0 = time.perf_counter()
    eng.forward(x, True, None, None, None, step, outputs=())
    t1 = time.perf_counter()
    eng.backward(1000.0, 10.0)
    t2 = time.perf_counter()
    eng.apply(1e-3, 1.0)
    t3 = time.perf_counter()
    tf += t1 - t0; tb += t2 - t1; ta += t3 - t2
t_host = time.perf_counter() - t_all0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t_all0
cap, eager = eng.graph_stats()
print("%s B=%d %s: host per step: forward %.0f us, backward %.0f us, apply %.0f us, sum %.0f us; loop without sync %.0f us/step; "
      "with final sync %.0f us/step; graphs captured %d, eager fallbacks %d"
      % (name, B, act, 1e6 * tf / N, 1e6 * tb / N, 1e6 * ta / N, 1e6 * (tf + tb + ta) / N, 1e6 * t_host / N, 1e6 * t_wall / N, cap, eager))
