#!/usr/bin/env python3
"""Which saved forward tensors change between the end of the forward and the end of the backward? (none may)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.common import COMPILE, engine_args, make_inputs
from multiscale_variational_autoencoder_amd.engine import Engine
name, B, dt = sys.argv[1], int(sys.argv[2]), sys.argv[3]
io = make_inputs(name, B)
eng = Engine(**engine_args(name, B), act_dtype=dt).bind()
eng.set_params(io["params"]); eng.set_state(io["state"])
d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=())
names = []
for k in eng.param_table:
    if k.endswith(".mn.conv0.w"):
        p = k[:-len(".conv0.w")]
        names += [p + ".t0", p + ".t1", p + ".out", p + ".gap", p + ".g", p + ".s0", p + ".ulin"]
    if k.endswith(".conv.w") or k.endswith(".convT.w") or k.endswith(".dense.w") or k.endswith(".conv_base.w"):
        names.append(k[:-2])
before = {n: eng.tensor(n, B).clone() for n in names}
eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
eng.sync()
bad = 0
for n in names:
    a, b = before[n].contiguous().view(-1).view(eng.torch.int32), eng.tensor(n, B).contiguous().view(-1).view(eng.torch.int32)
    nd = int((a != b).sum())
    if nd:
        bad += 1
        idx = (a != b).nonzero().reshape(-1)
        print("CHANGED %-24s %d of %d elements, first flat index %d, last %d" % (n, nd, a.numel(), int(idx[0]), int(idx[-1])))
print("tensors changed:", bad, "of", len(names))
