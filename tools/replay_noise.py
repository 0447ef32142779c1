import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from tests.common import CONFIGS, engine_args
from multiscale_variational_autoencoder_amd.engine import Engine
from multiscale_variational_autoencoder_amd.initializers import init_params
name, B = "c32nb", 16
x = np.random.default_rng(5).uniform(0, 255, (B,) + tuple(CONFIGS[name]["input_dims"])).astype(np.float32)
runs = []
for graphs, streams in (("0", "0"), ("0", "0"), ("0", "0"), ("1", "1"), ("1", "1"), ("1", "1")):
    os.environ["MVAE_GRAPHS"] = graphs; os.environ["MVAE_STREAMS"] = streams
    eng = Engine(**engine_args(name, B)).bind()
    eng.set_params(init_params(eng.param_table, 42))
    xd = eng.to_device(x)
    for step in range(4):
        eng.train_step(xd, 1e-3, 1000.0, 10.0, 1.0, seed=100 + step)
    p = eng.get_params()
    runs.append(np.concatenate([np.asarray(p[k], np.float64).ravel() for k in p]))
    eng.close()
d = lambda i, j: np.linalg.norm(runs[i] - runs[j])
print("eager-eager: %.5f %.5f %.5f   graph-graph: %.5f %.5f %.5f   eager-graph: %s" % (d(0, 1), d(0, 2), d(1, 2), d(3, 4), d(3, 5), d(4, 5),
      " ".join("%.5f" % d(i, j) for i in range(3) for j in range(3, 6))))
