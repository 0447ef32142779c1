"""Median step time of the C32-nb train step at a given batch and activation dtype: python tools/batch_time.py B [f32|bf16]"""
import sys, os, json
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from bench import NB
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params
B = int(sys.argv[1])
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
eng = Engine((32, 32, 3), [16] * 3, NB, NB, 0.0, 255.0, 0.01, B, act_dtype=dt).bind(0)
eng.set_params(init_params(eng.param_table, 42))
x = eng.to_device(np.random.default_rng(1).uniform(0, 255, (B, 32, 32, 3)))
for i in range(5):
    eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=i)
steps = 60
evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
torch.cuda.synchronize()
evs[0].record(eng.stream)
for i in range(steps):
    eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=10 + i)
    evs[i + 1].record(eng.stream)
torch.cuda.synchronize()
per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
print(B, dt, float(np.median(per)), "ms/step", B / float(np.median(per)) * 1e3, "images/s")
