"""Where do the split-bf16 and float32-MFMA 5x5 kernels disagree?  (debug aid for kernels_split.hip)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import COMPILE, engine_args, make_inputs, oracle_config
from multiscale_variational_autoencoder_amd.engine import Engine

name, B = sys.argv[1], int(sys.argv[2])
io = make_inputs(name, B)
res = {}
for split in (1, 0):
    os.environ["MVAE_SPLIT_CONV"] = str(split)
    eng = Engine(**engine_args(name, B)).bind(0)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=())
    if len(sys.argv) > 3:
        eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    L = len(oracle_config(name).z_dims)
    for s in range(L):
        for nm in ("enc%d.b0.conv" % s, "dec%d.b0.convT" % s, "dec%d.dense" % s, "enc%d.conv_base" % s):
            res[(split, nm)] = eng.tensor(nm, B).cpu().numpy().astype(np.float64)
    eng.close()
H0 = oracle_config(name).input_dims[0]
for s in range(L):
    for nm in ("enc%d.conv_base" % s, "dec%d.dense" % s, "enc%d.b0.conv" % s, "dec%d.b0.convT" % s):
        a, b = res[(1, nm)], res[(0, nm)]
        dd = np.abs(a - b)
        print(nm, "rel %.3e max %.3e of %.3e" % (np.linalg.norm(a - b) / np.linalg.norm(b), dd.max(), np.abs(b).max()), "nbad", int((dd > 1e-5 * np.abs(b).max()).sum()), "of", dd.size)
        if dd.max() > 1e-5 * np.abs(b).max() and "conv" in nm and "base" not in nm:
            C = 64
            Hs = (H0 >> s) if "convT" in nm else (H0 >> s) // 2
            d4 = dd.reshape(B, Hs, Hs, C)
            bad = np.argwhere(d4 > 1e-5 * np.abs(b).max())
            print("  bad b", np.unique(bad[:, 0])[:10], "y", np.unique(bad[:, 1])[:40], "x", np.unique(bad[:, 2])[:40], "c", np.unique(bad[:, 3])[:70])
            y, x = bad[0][1], bad[0][2]
            print("  first bad", bad[0], a.reshape(B, Hs, Hs, C)[tuple(bad[0])], b.reshape(B, Hs, Hs, C)[tuple(bad[0])])
