#!/usr/bin/env python3
"""Locate a config-dependent gradient error (see tools/grad_outlier_ab.py): run c64nb B=2 once per environment
variant, keep every gradient and the saved tensors of the suspected block, and compare variant vs variant vs oracle."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")
BLK = "dec0.b3.mn"


def child(tag, name, B):
    from tests.common import COMPILE, engine_args, make_inputs
    from multiscale_variational_autoencoder_amd.engine import Engine
    io = make_inputs(name, B)
    eng = Engine(**engine_args(name, B)).bind()
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("losses",))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    g = eng.get_grads()
    sv = {}
    for t in ("t0", "t1", "out", "g", "gap"):
        sv["act/" + t] = eng.tensor(BLK + "." + t, B).cpu().numpy()
    sv["act/prev_out"] = eng.tensor("dec0.b2.mn.out", B).cpu().numpy()
    np.savez(os.path.join(OUT, "locate_%s.npz" % tag), **{"g/" + k: v for k, v in g.items()}, **sv)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    name, B = "c64nb", 2
    os.makedirs(OUT, exist_ok=True)
    variants = (("default", {}), ("s0", {"MVAE_STREAMS": "0"}), ("s0b", {"MVAE_STREAMS": "0"}), ("lsb0", {"MVAE_LSB_MASK": "0"}))
    for tag, env in variants:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", tag, name, str(B)], env=e, capture_output=True, text=True)
        print(tag, "rc", r.returncode, r.stderr[-300:] if r.returncode else "", flush=True)
    from tests.common import COMPILE, make_inputs, oracle_config, reg_grad
    from oracle.mvae_oracle import Oracle, param_table
    io = make_inputs(name, B)
    inter = {}
    res, G = Oracle(oracle_config(name)).loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                                        COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], inter=inter)
    P, _ = param_table(oracle_config(name))
    rg = reg_grad(io["params"], {k: dict(reg=v[1]) for k, v in P.items()})
    runs = {tag: np.load(os.path.join(OUT, "locate_%s.npz" % tag)) for tag, _ in variants}
    rep = {}
    for tag, f in runs.items():
        errs = {}
        for k in G:
            ref = np.asarray(G[k], np.float64)
            got = f["g/" + k].astype(np.float64) + rg[k]
            n = np.linalg.norm(ref)
            if n > 1e-3:
                errs[k] = float(np.linalg.norm(got - ref) / n)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:25]
        rep[tag] = top
        print("====", tag)
        for k, v in top:
            print("   %-28s %.3e" % (k, v))
        blk = {k: v for k, v in errs.items() if k.startswith(BLK)}
        print("   block:", json.dumps(blk))
        # per-channel error of the conv0 bias gradient
        kb = BLK + ".conv0.b"
        ref = np.asarray(G[kb], np.float64); got = f["g/" + kb].astype(np.float64)
        pe = np.abs(got - ref) / np.abs(ref).mean()
        print("   conv0.b per-channel err/mean|ref|: max %.3e at ch %d, median %.3e, top5 %s" % (pe.max(), pe.argmax(), np.median(pe), np.sort(pe)[-5:]))
        # ReLU mask agreement of t0 with the oracle
        t0 = f["act/t0"].reshape(B, 64, 64, 64)
        for key in inter:
            if key.startswith(BLK) and key.endswith("t0"):
                o = np.transpose(inter[key].detach().numpy(), (0, 2, 3, 1))
                flips = int(((t0 > 0) != (o > 0)).sum())
                print("   t0 mask flips vs oracle:", flips, "of", t0.size, " max|t0-o|", float(np.abs(t0 - o).max()))
    a, b = runs["default"], runs["s0"]
    for t in ("t0", "t1", "out", "g", "gap", "prev_out"):
        d = np.abs(a["act/" + t] - b["act/" + t]).max()
        print("act", t, "default vs s0 max abs diff", float(d))
    for k in (BLK + ".conv0.w", BLK + ".conv0.b", BLK + ".dw.w", BLK + ".conv2.w"):
        d = np.abs(a["g/" + k] - b["g/" + k])
        print("grad", k, "default vs s0: max", float(d.max()), "rel", float(np.linalg.norm(a["g/" + k] - b["g/" + k]) / np.linalg.norm(a["g/" + k])),
              " s0 vs s0b rel", float(np.linalg.norm(runs["s0b"]["g/" + k] - b["g/" + k]) / np.linalg.norm(a["g/" + k])))
    with open(os.path.join(OUT, "grad_locate.json"), "w") as fjs:
        json.dump(rep, fjs, indent=1)
