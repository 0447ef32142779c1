"""How much of the C32-nb step is the scale-0 chain alone?  Times the train step of (a) the 3-scale workload, (b) a
1-scale model of the same blocks at 32x32 (scale 0's chain + the serial pre/post kernels), (c) 1-scale at 16x16 and
8x8 inputs (what scales 1 and 2 cost on their own)."""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import NB
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params

def run(dims, levels, B=512, steps=60):
    eng = Engine(dims, [16] * levels, NB, NB, 0.0, 255.0, 0.01, B).bind(0)
    eng.set_params(init_params(eng.param_table, 42))
    x = eng.to_device(np.random.default_rng(1).uniform(0, 255, (B,) + tuple(dims)))
    for i in range(5):
        eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=i)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    evs[0].record(eng.stream)
    for i in range(steps):
        eng.train_step(x, 1e-3, 1000.0, 10.0, 1.0, seed=10 + i)
        evs[i + 1].record(eng.stream)
    torch.cuda.synchronize()
    per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    print("graphs (captured, eager fallbacks):", eng.graph_stats(), file=sys.stderr)
    eng.close()
    return float(np.median(per))

if len(sys.argv) > 2:
    print(run((int(sys.argv[1]),) * 2 + (3,), int(sys.argv[2]), steps=10)); sys.exit(0)
out = {"3 scales 32x32": run((32, 32, 3), 3), "2 scales 32x32": run((32, 32, 3), 2),
       "2 scales 16x16": run((16, 16, 3), 2), "2 scales 8x8": run((8, 8, 3), 2)}
print(json.dumps(out))
