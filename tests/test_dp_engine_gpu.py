"""GPU, world_size 2: the ENGINE's data-parallel step (Engine.train_step: forward + backward on the rank's shard, ONE
all-reduce of the [gradients | BN batch statistics | metrics] arena, Adagrad with grad_scale = 1/N) against the oracle
evaluated with per-replica BatchNorm groups.  Both ranks share cuda:0 and the collective goes through
gloo -- the one-GPU rehearsal of what the driver runs with backend "nccl" (= RCCL), one rank per GPU."""
import os
import socket

import numpy as np
import pytest

from tests.common import COMPILE, engine_args, make_inputs, oracle_config, rel_err, structurally_zero

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, name, B, overlap="0", act="f32"):
    os.environ["MVAE_DP_OVERLAP"] = overlap
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multiscale_variational_autoencoder_amd.engine import Engine
    io = make_inputs(name, B)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    eng = Engine(**engine_args(name, per), act_dtype=act).bind(0)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k][sl]) for k in ("x", "eps", "noise", "keep")}
    eng.train_step(d["x"], COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                   COMPILE["clip_norm"], eps=d["eps"], noise=d["noise"], keep_mask=d["keep"])
    eng.sync()
    m = eng.metrics()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), count=m["count"], r_exp=m["r_exp"], kl=m["vae_kl_loss"],
             **{"p/" + k: v for k, v in eng.get_params().items()}, **{"a/" + k: v for k, v in eng.get_accum().items()},
             **{"s/" + k: v for k, v in eng.get_state().items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,B,overlap", [("tiny", 8, "0"), ("c32nb", 8, "0"), ("c32nb", 8, "force")])
def test_engine_dp2_matches_oracle_with_per_replica_batchnorm(tmp_path, name, B, overlap):
    """overlap = "force": the two-phase backward with the Dense-weight region all-reduced while phase 1 runs."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), name, B, overlap), nprocs=world, join=True)
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    p0 = {k: np.asarray(v, np.float64) for k, v in io["params"].items()}
    a0 = {k: np.full(v.shape, 0.1) for k, v in p0.items()}
    st0 = {k: np.asarray(v, np.float64) for k, v in io["state"].items()}
    res, G, p1, a1, st1 = orc.train_step(p0, a0, st0, io["x"], io["eps"], io["noise"], io["keep"],
                                         COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                                         COMPILE["clip_norm"], bn_group_size=B // world)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    zero = structurally_zero(G)
    lr = COMPILE["learning_rate"]
    worst_p = worst_a = 0.0
    for k in p1:
        # replicas apply the same update to the same all-reduced gradients: they stay identical
        assert np.array_equal(r0["p/" + k], r1["p/" + k]), k
        w = 0.1 if k in zero else 1.0                     # noise-only gradients (see test_adagrad_trajectory_parity)
        worst_p = max(worst_p, w * float(np.abs(r0["p/" + k] - p1[k]).max() / lr))
        worst_a = max(worst_a, (0.05 if k in zero else 1.0) * rel_err(r0["a/" + k], a1[k]))
    # accumulators hold g^2: twice the gradient's relative error; BatchNorm over 4-image replicas amplifies fp32 noise
    assert worst_p <= 3e-2 and worst_a <= 2e-2, (worst_p, worst_a)
    for k, v in st1.items():                              # BN moving statistics: mean over the two replicas' batch stats
        assert rel_err(r0["s/" + k], np.asarray(v)) <= 1e-5, k
    assert float(r0["count"]) == B                        # metrics block is part of the same all-reduce
    assert abs(float(r0["r_exp"]) - float(np.mean(res["r_exp"]))) <= 1e-4 * abs(float(np.mean(res["r_exp"])))
    assert abs(float(r0["kl"]) - float(np.mean(res["kl"]))) <= 1e-4 * abs(float(np.mean(res["kl"]))) + 1e-6



@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", ["0", "force"])
def test_engine_dp2_bf16_c64nb(tmp_path, overlap):
    """BASELINE config 5's combination (bf16 activations + data parallel; here the 64x64 / 5-scale shrink of C256-nb, two
    ranks of 8 images sharing one GPU over gloo, single-message and two-phase exchange).  What data parallelism adds is
    exact -- the all-reduce sums float32 arenas -- so the replicas must stay bit-identical; against the float64 oracle
    with per-replica BatchNorm groups the bars are the bf16 class of tests/test_bf16_gpu.py (a 4-image replica: its
    small-batch bars): per tensor, the relative error of the applied UPDATE (Adagrad's first step is lr * g / sqrt(0.1 + g^2):
    at most the gradient's relative error).  Measured (8 images per replica): median 3.3e-2 .. 3.5e-2, 90th percentile 9e-2,
    worst weight 0.24 (a squeeze-excite Dense behind a BatchNorm over 8 rows: the same tensors lead the single-GPU bf16
    report at small batches), metrics 6e-4, BatchNorm moving statistics 1e-5.  Bars: median <= 7e-2, worst weight <= 0.5,
    metrics <= 2e-2, statistics <= 2e-2 -- a dropped or doubled shard moves every one of them by O(1)."""
    import json
    import torch.multiprocessing as mp
    name, B, world = "c64nb", 16, 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), name, B, overlap, "bf16"), nprocs=world, join=True)
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    p0 = {k: np.asarray(v, np.float64) for k, v in io["params"].items()}
    a0 = {k: np.full(v.shape, 0.1) for k, v in p0.items()}
    st0 = {k: np.asarray(v, np.float64) for k, v in io["state"].items()}
    res, G, p1, a1, st1 = orc.train_step(p0, a0, st0, io["x"], io["eps"], io["noise"], io["keep"],
                                         COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                                         COMPILE["clip_norm"], bn_group_size=B // world)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    zero = structurally_zero(G)
    errs, worst_w, worst_v = [], ("", 0.0), ("", 0.0)
    for k in p1:
        assert np.array_equal(r0["p/" + k], r1["p/" + k]), k            # replicas stay identical
        assert np.array_equal(r0["a/" + k], r1["a/" + k]), k
        if k in zero:
            continue
        step = p1[k] - p0[k]
        e = float(np.linalg.norm((r0["p/" + k] - p1[k]).ravel()) / max(np.linalg.norm(step.ravel()), 1e-30))
        errs.append(e)
        if p1[k].ndim >= 2:
            worst_w = max(worst_w, (k, e), key=lambda kv: kv[1])
        else:
            worst_v = max(worst_v, (k, e), key=lambda kv: kv[1])
    st_err = max(rel_err(r0["s/" + k], np.asarray(v)) for k, v in st1.items())
    rep = {"median": float(np.median(errs)), "p90": float(np.percentile(errs, 90)), "worst_weight": worst_w,
           "worst_vector": worst_v, "state": st_err,
           "r_exp": abs(float(r0["r_exp"]) / float(np.mean(res["r_exp"])) - 1.0),
           "kl": abs(float(r0["kl"]) / float(np.mean(res["kl"])) - 1.0)}
    from tests.common import ROOT
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dp2_bf16_c64nb_overlap%s.json" % overlap), "w") as f:
        json.dump(rep, f, indent=1)
    assert float(r0["count"]) == B
    print(json.dumps(rep))
    assert rep["median"] <= 7e-2 and rep["worst_weight"][1] <= 0.5, rep
    assert rep["r_exp"] <= 2e-2 and rep["kl"] <= 2e-2 and rep["state"] <= 2e-2, rep
