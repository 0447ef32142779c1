"""Hunt for an intermittent corruption of dec*.dense: compare d0 on the device against z.W+b computed on the host from the
device's own z, repeatedly, in both conv modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.common import COMPILE, engine_args, make_inputs, oracle_config
from multiscale_variational_autoencoder_amd.engine import Engine
name, B, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bwd = len(sys.argv) > 4
io = make_inputs(name, B)
L = len(oracle_config(name).z_dims)
for split in (1, 0):
    os.environ["MVAE_SPLIT_CONV"] = str(split)
    eng = Engine(**engine_args(name, B)).bind(0)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    nbad = 0
    for rep in range(reps):
        eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=())
        if bwd:
            eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
        for s in range(L):
            z = eng.tensor("enc%d.z" % s, B).cpu().numpy().astype(np.float64)
            d0 = eng.tensor("dec%d.dense" % s, B).cpu().numpy().astype(np.float64)
            ref = z @ io["params"]["dec%d.dense.w" % s].astype(np.float64) + io["params"]["dec%d.dense.b" % s].astype(np.float64)
            bad = np.argwhere(np.abs(d0 - ref) > 1e-5)
            if len(bad):
                nbad += 1
                print("split", split, "rep", rep, "scale", s, "bad", len(bad), "rows", np.unique(bad[:, 0]), "cols", bad[:, 1].min(), bad[:, 1].max(),
                      "of", d0.shape[1], "maxerr", np.abs(d0 - ref).max())
                for (r_, c_) in bad[:6]:
                    print("    col", c_, "got", np.float32(d0[r_, c_]).view(np.uint32).item().__format__("08x"), "ref", np.float32(ref[r_, c_]).view(np.uint32).item().__format__("08x"), d0[r_, c_], ref[r_, c_])
                print("    cols", bad[:, 1][:32])
    print("split", split, "reps", reps, "bad events", nbad)
    eng.close()
