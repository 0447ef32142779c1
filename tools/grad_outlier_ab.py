#!/usr/bin/env python3
"""A/B of the gradient error against the fp64 oracle (dec0.b3.mn.conv0.w at 1.7e-3 in c64nb):
default build vs MVAE_LSB_MASK=0 (depthwise backward reads t1 instead of the mantissa-LSB mask) vs MVAE_GRAD_SLOTS=0
(no slot copies: one atomic target) vs both.  Prints the worst tensors of each variant; one process per variant because
the switches are read at bind time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(name, B):
    import numpy as np
    from tests.common import COMPILE, engine_args, grad_errors, make_inputs, oracle_config, reg_grad
    from oracle.mvae_oracle import Oracle
    from multiscale_variational_autoencoder_amd.engine import Engine
    io = make_inputs(name, B)
    res, G = Oracle(oracle_config(name)).loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"],
                                                        io["keep"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    eng = Engine(**engine_args(name, B)).bind()
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("losses",))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    g = eng.get_grads()
    rg = reg_grad(io["params"], eng.param_table)
    # plain relative error per tensor (no floor) next to the test's criterion
    rel = {}
    for k in G:
        ref = np.asarray(G[k], np.float64)
        got = g[k].astype(np.float64) + rg[k]
        rel[k] = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    crit = grad_errors({k: g[k].astype(np.float64) + rg[k] for k in G}, G)
    worst = sorted(crit.items(), key=lambda kv: -kv[1])[:8]
    print(json.dumps({"worst_criterion": worst, "plain_rel_of_worst": {k: rel[k] for k, _ in worst},
                      "norm_of_worst": {k: float(np.linalg.norm(np.asarray(G[k], np.float64))) for k, _ in worst}}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "c64nb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    out = {}
    for tag, env in (("default", {}), ("lsb_off", {"MVAE_LSB_MASK": "0"}), ("slots_off", {"MVAE_GRAD_SLOTS": "0"}),
                     ("both_off", {"MVAE_LSB_MASK": "0", "MVAE_GRAD_SLOTS": "0"}),
                     ("eager_1stream", {"MVAE_GRAPHS": "0", "MVAE_STREAMS": "0"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, __file__, "child", name, str(B)], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        out[tag] = json.loads(line[-1]) if line else {"error": r.stderr[-2000:]}
        print(tag, json.dumps(out[tag]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "grad_outlier_ab_%s.json" % name), "w") as f:
        json.dump(out, f, indent=1)
