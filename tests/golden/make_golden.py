"""Generates tests/golden/tiny_case.npz from the CPU oracle (the reference itself cannot run: SURVEY.md 8(c)).
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.mvae_oracle import Oracle                     # noqa: E402
from tests.common import COMPILE, make_inputs, oracle_config   # noqa: E402


def main():
    io = make_inputs("tiny", 4)
    orc = Oracle(oracle_config("tiny"))
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    from tests.common import reg_grad
    from oracle.mvae_oracle import param_table
    P, _ = param_table(oracle_config("tiny"))
    rg = reg_grad(io["params"], P)
    blob = dict(x=io["x"], eps=io["eps"], noise=io["noise"], keep=io["keep"], recon=res["recon"].astype(np.float32),
                losses=np.stack([res["r"], res["r_exp"], res["kl"]], 1), loss=np.float64(res["loss"]),
                data_loss=np.float64(res["data_loss"]), reg_loss=np.float64(res["reg_loss"]))
    for k, v in io["params"].items():
        blob["p/" + k] = v
        blob["g/" + k] = (G[k] - rg[k]).astype(np.float32)      # data-loss gradient (what mvae_backward returns)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tiny_case.npz"), **blob)
    print("wrote tiny_case.npz: loss %.6f data %.6f reg %.6f" % (res["loss"], res["data_loss"], res["reg_loss"]))


if __name__ == "__main__":
    main()
