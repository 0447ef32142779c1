"""CPU, world_size 2 over gloo: the data-parallel recipe of SURVEY.md 8(e) / Engine.train_step -- shard the batch,
ONE all-reduce(sum) of the flat [gradients | BN batch statistics | metrics] arena, scale by 1/N, then identical
per-variable clipnorm + Adagrad on every rank -- reproduces the single-process oracle evaluated with per-replica
BatchNorm groups (bn_group_size = B/N).  The per-rank "device" here is the oracle itself (no GPU in this suite);
the GPU engine issues exactly the same collective on the same arena layout (engine.py:train_step)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import COMPILE, make_inputs, oracle_config


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.mvae_oracle import Oracle
    name, B = "tiny", 8
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"][sl], io["eps"][sl], io["noise"][sl], io["keep"][sl],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    keys = list(G)
    skeys = list(res["new_state"])
    # the reduce arena: [grads | BN statistics (here: the updated moving stats, linear in the batch stats) | metrics]
    arena = np.concatenate([G[k].ravel() for k in keys] + [np.asarray(res["new_state"][k]).ravel() for k in skeys] +
                           [np.array([per, res["r"].sum(), res["r_exp"].sum(), res["kl"].sum()])])
    t = torch.from_numpy(arena.copy())
    dist.all_reduce(t)                                   # ONE collective per step
    t /= world
    red = t.numpy()
    off = 0
    Gr = {}
    for k in keys:
        n = G[k].size
        Gr[k] = red[off:off + n].reshape(G[k].shape); off += n
    st = {}
    for k in skeys:
        n = np.asarray(res["new_state"][k]).size
        st[k] = red[off:off + n]; off += n
    a0 = {k: np.full(np.shape(v), 0.1) for k, v in io["params"].items()}
    p1, a1 = orc.adagrad_step(io["params"], a0, Gr, COMPILE["learning_rate"], COMPILE["clip_norm"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), metrics=red[off:off + 4] * world,
             **{"p/" + k: v for k, v in p1.items()}, **{"s/" + k: v for k, v in st.items()},
             **{"g/" + k: v for k, v in Gr.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_dp2_equals_oracle_with_per_replica_batchnorm(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from oracle.mvae_oracle import Oracle
    name, B = "tiny", 8
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], bn_group_size=B // world)
    a0 = {k: np.full(np.shape(v), 0.1) for k, v in io["params"].items()}
    p1, _ = orc.adagrad_step(io["params"], a0, G, COMPILE["learning_rate"], COMPILE["clip_norm"])
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    gmax = max(np.abs(g).max() for g in G.values())
    for k in G:
        assert np.array_equal(r0["g/" + k], r1["g/" + k])                     # replicas stay identical
        assert np.allclose(r0["g/" + k], G[k], rtol=1e-9, atol=1e-12 * gmax), k   # == oracle with BN groups of B/N
        assert np.allclose(r0["p/" + k], p1[k], rtol=1e-12, atol=1e-14), k
    for k, v in res["new_state"].items():
        assert np.allclose(r0["s/" + k], np.asarray(v).ravel(), rtol=1e-9, atol=1e-12), k
    assert np.allclose(r0["metrics"], [B, res["r"].sum(), res["r_exp"].sum(), res["kl"].sum()], rtol=1e-10)
    # and it is NOT the single-replica (full-batch BatchNorm) result: the SURVEY 8(e) caveat is real
    _, Gfull = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                  COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    assert max(np.abs(Gfull[k] - G[k]).max() for k in G) > 1e-6
