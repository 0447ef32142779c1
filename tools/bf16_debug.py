#!/usr/bin/env python3
"""bf16 bring-up: run the same injected step on a float32 engine and a bfloat16 engine and print, in graph order, the
relative difference of every saved forward tensor and of every gradient tensor (bf16 rounding gives ~1e-3..1e-2; a wrong
kernel gives O(1) at the first tensor it produces)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c64nb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from tests.common import COMPILE, engine_args, make_inputs, rel_err
    from multiscale_variational_autoencoder_amd.engine import Engine
    io = make_inputs(name, B)
    runs = {}
    for dt in ("f32", "bf16"):
        eng = Engine(**engine_args(name, B), act_dtype=dt).bind()
        eng.set_params(io["params"]); eng.set_state(io["state"])
        d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
        out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "losses", "mu", "log_var"))
        fw = {}
        for k in eng.param_table:
            if k.endswith(".conv_base.w"):
                fw[k[:-2]] = eng.tensor(k[:-2], B).cpu().numpy()
            if k.endswith(".mn.conv0.w"):
                p = k[:-len(".conv0.w")]
                for t in ("t0", "t1", "gap", "g", "out"):
                    fw[p + "." + t] = eng.tensor(p + "." + t, B).cpu().numpy()
            if k.endswith(".conv.w") or k.endswith(".convT.w"):
                fw[k[:-2]] = eng.tensor(k[:-2], B).cpu().numpy()
            if k.endswith(".dense.w"):
                fw[k[:-2]] = eng.tensor(k[:-2], B).cpu().numpy()
            if k.endswith(".out.w"):
                fw[k[:-6] + ".y"] = eng.tensor(k[:-6] + ".y", B).cpu().numpy()
        fw["mu"] = out["mu"].cpu().numpy(); fw["log_var"] = out["log_var"].cpu().numpy()
        fw["recon"] = out["recon"].cpu().numpy(); fw["losses"] = out["losses"].cpu().numpy()
        eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
        g = eng.get_grads()
        runs[dt] = (fw, g, eng.scale_dtypes())
        eng.close()
    print("scale dtypes (bf16 engine):", runs["bf16"][2])
    fa, ga, _ = runs["f32"]
    fb, gb, _ = runs["bf16"]
    print("---- forward (rel diff bf16 vs f32)")
    for k in fa:
        e = rel_err(fb[k], fa[k])
        print("%-28s %.3e%s" % (k, e, "   <<<<" if e > 0.05 else ""))
    print("---- gradients (rel diff bf16 vs f32), backward order")
    keys = list(ga)
    for k in reversed(keys):
        n = np.linalg.norm(ga[k])
        e = rel_err(gb[k], ga[k])
        flag = "   <<<<" if (e > 0.1 and n > 1e-3) else ""
        print("%-28s %.3e  |g|=%.3e%s" % (k, e, n, flag))


if __name__ == "__main__":
    main()
