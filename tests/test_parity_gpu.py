"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical inputs, weights
and injected epsilon / noise / dropout masks.  Tolerances (fp32 kernels vs float64 oracle), BASELINE.md section 5:
ELBO terms <= 1e-4 relative, reconstructions <= 1e-4 * 255 absolute, gradients per tensor
||got-ref|| <= 2e-4 * max(||ref||, 0.1 * rms * sqrt(n)) (tests/common.py:grad_errors; the second arm is the floor for
small tensors) and 2e-3 for the tensors whose true gradient is identically zero (a bias that feeds straight into
BatchNorm: pure fp32 cancellation noise).

The gradient is compared along the device's own branch at every kink (Oracle.set_kink_masks): the device's ReLU /
hard-sigmoid active sets are read back and the oracle differentiates with them.  Round 1 left an unexplained 1.7e-3
outlier (dec0.b3.mn.conv0.w, c64nb): tools/grad_stress.py shows ONE ReLU input of 2^-28 in that block flipping with the
summation order of the forward's float atomics (15 % of the runs), at r_loss_factor = 1000 and batch 2 a single
pixel-channel moves that 64x64 weight gradient by 1.7e-3.  The number of units that sit on the other side of a kink
than in the oracle's own float64 forward, and their distance from the kink, are bounded (check_kink_report)."""
import json
import os

import numpy as np
import pytest

from tests.common import (COMPILE, CONFIGS, ROOT, check_kink_report, device_kink_masks, engine_args, grad_errors,
                          make_inputs, oracle_config, reg_grad, rel_err, structurally_zero)

pytestmark = pytest.mark.gpu

TOL_ELBO = 1e-4
TOL_RECON_ABS = 1e-4 * 255.0
TOL_GRAD = 2e-4            # every tensor whose true gradient is not identically zero
TOL_GRAD_ZERO = 2e-3       # structurally zero gradients: fp32 cancellation noise over the floor 0.1 * rms * sqrt(n)
OUT = os.path.join(ROOT, "gpurun_out")


def _engine(name, B):
    from multiscale_variational_autoencoder_amd.engine import Engine
    return Engine(**engine_args(name, B)).bind()


def _run_pair(name, B, seed=0):
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B, seed)
    oc = oracle_config(name)
    orc = Oracle(oc)
    inter = {}
    eng = _engine(name, B)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "mu", "log_var", "z", "losses"))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    eng.torch.cuda.synchronize()
    orc.set_kink_masks(device_kink_masks(eng, B))          # differentiate along the device's branch at every kink
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], inter=inter)
    return io, oc, orc, res, G, inter, eng, out


def _dump(name, report):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_%s.json" % name), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)


def parity_case(name, B, tag=None, env_check=None):
    """One forward + backward of config `name` at batch B on the device against the oracle: every saved tensor, the loss
    terms, the reconstruction, every gradient tensor, the kink report.  Also the body of the subprocess cases below
    (`python -m tests.test_parity_gpu NAME B TAG`), which run it under launch-geometry switches that are read once per process."""
    io, oc, orc, res, G, inter, eng, out = _run_pair(name, B)
    if env_check:
        env_check(eng)
    rep = {}
    # ---- saved intermediates, layer by layer (first failing layer localises a kernel bug)
    for k, v in inter.items():
        try:
            t = eng.tensor(k, B).cpu().numpy()
        except KeyError:
            continue
        ref = v.detach().numpy()
        if ref.ndim == 4:
            ref = np.transpose(ref, (0, 2, 3, 1))
        rep["fwd/" + k] = rel_err(t.reshape(ref.shape), ref)
    losses = out["losses"].cpu().numpy().astype(np.float64)
    rep["loss/r"] = rel_err(losses[:, 0], res["r"]); rep["loss/r_exp"] = rel_err(losses[:, 1], res["r_exp"])
    rep["loss/kl"] = rel_err(losses[:, 2], res["kl"])
    for s in range(oc.levels):
        rep["loss/kl_scale%d" % s] = rel_err(losses[:, 3 + s], res["kl_scale"][:, s])
    elbo_gpu = (COMPILE["r_loss_factor"] * losses[:, 1] + COMPILE["kl_loss_factor"] * losses[:, 2]).mean()
    rep["loss/elbo_rel"] = abs(elbo_gpu - res["data_loss"]) / abs(res["data_loss"])
    rep["loss/reg_rel"] = abs(eng.reg_loss() - res["reg_loss"]) / max(abs(res["reg_loss"]), 1e-30)
    rep["recon_abs"] = float(np.abs(out["recon"].cpu().numpy() - res["recon"]).max())
    for k in ("mu", "log_var", "z"):
        rep["out/" + k] = rel_err(out[k].cpu().numpy(), res[k])
    grads = eng.get_grads()
    rg = reg_grad(io["params"], eng.param_table)
    gerr = grad_errors({k: grads[k].astype(np.float64) + rg[k] for k in G}, G)
    for k, e in gerr.items():
        rep["grad/" + k] = e
    worst = max(gerr.items(), key=lambda kv: kv[1])
    m = eng.metrics()
    rep["metrics/r"] = abs(m["vae_r_loss"] - res["r"].mean()) / abs(res["r"].mean())
    rep["metrics/kl"] = abs(m["vae_kl_loss"] - res["kl"].mean()) / abs(res["kl"].mean())
    _dump(tag or name, rep)
    bad_fwd = {k: v for k, v in rep.items() if k.startswith("fwd/") and v > 1e-4}
    assert not bad_fwd, bad_fwd
    assert rep["recon_abs"] <= TOL_RECON_ABS, rep["recon_abs"]
    for k in ("loss/r", "loss/r_exp", "loss/kl", "loss/elbo_rel", "loss/reg_rel", "metrics/r", "metrics/kl"):
        assert rep[k] <= TOL_ELBO, (k, rep[k])
    kr = orc.kink_report()
    rep["kink/units"], rep["kink/flips"], rep["kink/max_abs_at_flip"] = kr["units"], kr["flips"], kr["max_abs_at_flip"]
    _dump(tag or name, rep)
    check_kink_report(kr)
    zero = structurally_zero(G)
    # the structurally-zero tensors (biases feeding BatchNorm) hold pure summation noise, which grows with the number of
    # rows a bias gradient sums: 2x the bar from 16384 rows per tensor on (c96nb 2.1e-3, C256-nb below 3e-3)
    rows = B * CONFIGS[name]["input_dims"][0] * CONFIGS[name]["input_dims"][1]
    tol_zero = TOL_GRAD_ZERO * (2 if rows >= 16384 else 1)
    bad = {k: v for k, v in gerr.items() if v > (tol_zero if k in zero else TOL_GRAD)}
    assert not bad, (worst, len(bad), dict(list(bad.items())[:12]))
    return rep


@pytest.mark.parametrize("name,B", [("tiny", 4), ("odd", 3), ("c32def", 4), ("main", 4), ("c32nb", 8), ("c64nb", 2), ("nbodd", 3), ("c96nb", 3)])
def test_forward_backward_parity(name, B):
    parity_case(name, B)


# The image-resident fused kernels (k_dw_bwd_conv0_s, k_mn_fwd_chain_s: one 512-thread block per CU walks whole images,
# `for (b = blockIdx.x; b < B; b += gridDim.x)`, with a fetch stream that runs on into the block's NEXT image) launch
# min(B, CUs) blocks: at the oracle's batch sizes every block has one image and the cross-image branch is dead, at the
# headline's batch 512 every block walks 2 (32- / 16-wide maps) or 4 (8-wide) images.  MVAE_FUSED_CUS* (read once per
# process, hence the subprocess) caps the grid at 8 blocks, which puts the multi-image path under the oracle at an
# affordable batch: c32nb B=20 -> blocks with 3 and 2 images (uneven tail) on 32- / 16- / 8-wide maps, c64nb B=20 likewise on
# its 32 / 16 / 8 wide scales (plus 64- and 4-wide ones on the other kernels), B=33 -> 5 and 4 images.  Same bars as above.
# reference: layer_blocks.py:594-641 (mobilenetV3_block forward; its backward per SURVEY appendix C).
MULTI_IMAGE_ENV = {"MVAE_FUSED_CUS": "8", "MVAE_FUSED_CUS16": "8", "MVAE_FUSED_CUS8": "8"}


def _subprocess_case(name, B, tag, env):
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "tests.test_parity_gpu", name, str(B), tag], cwd=ROOT, env=e, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-6000:]
    return r.stdout


@pytest.mark.parametrize("name,B,det", [("c32nb", 20, "0"), ("c64nb", 20, "0"), ("c32nb", 33, "0"), ("c32nb", 20, "1")])
def test_multi_image_blocks_parity(name, B, det):
    env = dict(MULTI_IMAGE_ENV, MVAE_DETERMINISTIC=det)
    out = _subprocess_case(name, B, "multi_%s_b%d_det%s" % (name, B, det), env)
    assert "fused launches: fwd" in out and "images per block" in out, out[-2000:]


def test_two_pass_batchnorm_statistics_parity():
    """MVAE_BN_ONEPASS=0: the decoder BatchNorm's batch statistics as two passes (sums, then squared deviations) -- the form the
    default one-pass kernels (k_colstat4<2>; the conv2 kernel's fused sums on bf16 scales) replaced, kept as their fallback."""
    out = _subprocess_case("c32nb", 20, "twopass_bn_c32nb_b20", dict(MULTI_IMAGE_ENV, MVAE_BN_ONEPASS="0"))
    assert "ok twopass_bn_c32nb_b20" in out, out[-2000:]


if __name__ == "__main__":          # subprocess body of test_multi_image_blocks_parity / test_two_pass_batchnorm_statistics_parity
    import sys
    _name, _B, _tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]

    def _check(eng):
        # the case is only worth its time if the fused kernels really ran, with several images per block
        st = eng.fused_launch_stats()
        print("fused launches: fwd %d bwd %d, images per block up to %d" % (st["fwd"], st["bwd"], st["max_images_per_block"]))
        # (the deterministic mode keeps the separate forward launches: the fused forward's GAP order is fixed anyway,
        # but its squeeze-excite inputs are not -- kernels_fused_fwd.hip)
        det = os.environ.get("MVAE_DETERMINISTIC", "") == "1"
        assert (st["fwd"] > 0 or det) and st["bwd"] > 0 and st["max_images_per_block"] >= 2, st

    rep = parity_case(_name, _B, tag=_tag, env_check=_check)
    print("ok", _tag, "elbo_rel %.2e" % rep["loss/elbo_rel"], "kink flips", rep["kink/flips"])



@pytest.mark.parametrize("name,B", [("tiny", 4), ("c32nb", 4)])
def test_adagrad_trajectory_parity(name, B):
    """3 optimiser steps (per-variable clipnorm, Adagrad a0=0.1, BN moving statistics) against the oracle."""
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B)
    oc = oracle_config(name)
    orc = Oracle(oc)
    eng = _engine(name, B)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    p = {k: np.asarray(v, np.float64) for k, v in io["params"].items()}
    a = {k: np.full(v.shape, 0.1) for k, v in p.items()}
    st = {k: np.asarray(v, np.float64) for k, v in io["state"].items()}
    rep = {}
    for step in range(3):
        stp = make_inputs(name, B, seed=step)
        d = {k: eng.to_device(stp[k]) for k in ("x", "eps", "noise", "keep")}
        eng.train_step(d["x"], COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                       COMPILE["clip_norm"], eps=d["eps"], noise=d["noise"], keep_mask=d["keep"])
        res, G, p, a, st = orc.train_step(p, a, st, stp["x"], stp["eps"], stp["noise"], stp["keep"],
                                          COMPILE["learning_rate"], COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"],
                                          COMPILE["clip_norm"])
        gp, ga, gs = eng.get_params(), eng.get_accum(), eng.get_state()
        # parameter error measured against the size of the step taken so far (lr * steps), not the parameter norm
        scale = COMPILE["learning_rate"] * (step + 1)
        zero = structurally_zero(G)      # noise-only gradients there: Adagrad turns them into noise-sized steps
        perr = {k: float(np.abs(gp[k] - p[k]).max() / scale) * (0.1 if k in zero else 1.0) for k in p}
        aerr = {k: rel_err(ga[k], a[k]) * (0.05 if k in zero else 1.0) for k in a}
        rep["step%d/param_vs_step" % step] = max(perr.values())
        rep["step%d/accum" % step] = max(aerr.values())
        rep["step%d/state" % step] = max(rel_err(gs[k], st[k]) for k in st)
        rep["step%d/worst_param" % step] = sorted(perr.items(), key=lambda kv: -kv[1])[:6]
        rep["step%d/worst_accum" % step] = sorted(aerr.items(), key=lambda kv: -kv[1])[:6]
    _dump("traj_" + name, rep)
    for k, v in rep.items():
        if "worst" in k:
            continue
        tol = 3e-2 if "param_vs_step" in k else (1e-5 if "state" in k else 5e-3)
        assert v <= tol, (k, v, rep)


@pytest.mark.parametrize("name,B", [("tiny", 4), ("c32nb", 4)])
def test_inference_parity(name, B):
    """model_trainable / encoder / decoder .predict: no noise/dropout, BN moving statistics, sampling still on."""
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    ref = orc.predict(io["params"], io["state"], io["x"], io["eps"])
    eng = _engine(name, B)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    out = eng.forward(eng.to_device(io["x"]), False, eng.to_device(io["eps"]), outputs=("recon", "z", "mu"))
    assert np.abs(out["recon"].cpu().numpy() - ref["recon"]).max() <= TOL_RECON_ABS
    assert rel_err(out["z"].cpu().numpy(), ref["z"]) <= 1e-5
    T = orc.tensors(io["params"])
    import torch
    with torch.no_grad():
        dec = orc.decode(T, io["state"], ref["z"]).numpy()
    got = eng.decode(eng.to_device(ref["z"])).cpu().numpy()
    assert np.abs(got - dec).max() <= TOL_RECON_ABS
    assert np.abs(got - ref["recon"]).max() <= TOL_RECON_ABS       # decoder(encoder(x)) == trainable(x)


@pytest.mark.parametrize("name,B", [("tiny", 4), ("c32nb", 16)])
def test_graph_replay_and_streams_match_eager(name, B, monkeypatch):
    """The production path replays captured hipGraphs with one stream per scale; it must compute what the eager
    single-stream path computes (same device-RNG seeds, so identical noise / dropout / epsilon draws).
    Float atomics make even two identical eager runs differ (the notebook's r_loss_factor=1000 amplifies rounding
    noise into Adagrad steps on the biases that feed BatchNorm), so the bound is the measured eager-vs-eager
    distance: a graph + streams run may lie at most 4x as far from its nearest eager run as the three eager runs lie from each
    other (pair distances of one mode scatter by 4x; the noise is ~1 % of the distance travelled, and a kernel that misbehaves
    under concurrency moves the weights by a visible share of that distance)."""
    from multiscale_variational_autoencoder_amd.initializers import init_params
    x = np.random.default_rng(5).uniform(0, 255, (B,) + tuple(CONFIGS[name]["input_dims"])).astype(np.float32)
    runs = []
    # three eager single-stream runs, the forked-graph replay, the segmented replay (MVAE_GRAPH_SEGMENTS=1: linear graphs + events)
    for graphs, streams, segments in (("0", "0", "0"), ("0", "0", "0"), ("0", "0", "0"), ("1", "1", "0"), ("1", "1", "1")):
        monkeypatch.setenv("MVAE_GRAPHS", graphs)
        monkeypatch.setenv("MVAE_STREAMS", streams)
        monkeypatch.setenv("MVAE_GRAPH_SEGMENTS", segments)
        eng = _engine(name, B)
        eng.set_params(init_params(eng.param_table, 42))
        xd = eng.to_device(x)
        mets = []
        for step in range(4):       # step 0 captures, steps 1.. replay
            eng.train_step(xd, 1e-3, 1000.0, 10.0, 1.0, seed=100 + step)
            mets.append(eng.metrics())
        if graphs == "1":
            assert eng.graph_stats() == (3, 0)
        runs.append((eng.get_params(), mets))
    (pa, ma), (pb, mb), (pe, me) = runs[:3]
    p0 = init_params(eng.param_table, 42)
    flat = lambda d: np.concatenate([np.asarray(d[k], np.float64).ravel() for k in pa])
    fa, fb, fe = flat(pa), flat(pb), flat(pe)
    travelled = np.linalg.norm(fa - flat(p0))
    # the distance between two runs of the SAME mode scatters by 4x from pair to pair (tools/replay_noise.py: 0.0016 .. 0.0073
    # for eager-eager, graph-graph and eager-graph alike at c32nb B=16): the noise level is the largest of three eager pairs,
    # a graph run is measured against its nearest eager run
    noise = max(np.linalg.norm(fa - fb), np.linalg.norm(fa - fe), np.linalg.norm(fb - fe))
    for pc, mc in runs[3:]:
        fc = flat(pc)
        dist = min(np.linalg.norm(fc - fa), np.linalg.norm(fc - fb), np.linalg.norm(fc - fe))
        assert np.isfinite(fc).all()
        assert dist <= 4.0 * max(noise, 1e-3 * travelled), (dist, noise, travelled)
        assert dist <= 0.1 * travelled, (dist, travelled)
        for m1, m2, m4, m3 in zip(ma, mb, me, mc):
            for k in m1:
                floor = max(abs(m1[k] - m2[k]), abs(m1[k] - m4[k]), abs(m2[k] - m4[k]), 1e-4 * abs(m1[k]), 1e-9)
                assert min(abs(m3[k] - m1[k]), abs(m3[k] - m2[k]), abs(m3[k] - m4[k])) <= 8.0 * floor, (k, m1[k], m2[k], m4[k], m3[k])


def test_c256nb_full_size_parity_and_training():
    """BASELINE config 4's model (256x256x3, 7 scales, notebook blocks; fp32 here) at batch 2 against the oracle
    (ELBO terms + reconstruction + a sample of gradients), then a few optimiser steps at batch 8: finite and
    decreasing loss.  Exercises the 64-bit indexing / column-strip paths the 32x32 configs never reach."""
    from oracle.mvae_oracle import Oracle
    name, B = "c256nb", 2
    io = make_inputs(name, B)
    orc = Oracle(oracle_config(name))
    eng = _engine(name, 8)
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "losses"))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    orc.set_kink_masks(device_kink_masks(eng, B))
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    check_kink_report(orc.kink_report())
    losses = out["losses"].cpu().numpy().astype(np.float64)
    elbo = (COMPILE["r_loss_factor"] * losses[:, 1] + COMPILE["kl_loss_factor"] * losses[:, 2]).mean()
    assert abs(elbo - res["data_loss"]) / abs(res["data_loss"]) <= TOL_ELBO
    assert rel_err(losses[:, 3:], res["kl_scale"]) <= TOL_ELBO
    assert np.abs(out["recon"].cpu().numpy() - res["recon"]).max() <= TOL_RECON_ABS
    grads = eng.get_grads()
    rg = reg_grad(io["params"], eng.param_table)
    gerr = grad_errors({k: grads[k].astype(np.float64) + rg[k] for k in G}, G)
    zero = structurally_zero(G)        # biases feeding BatchNorm: pure fp32 cancellation noise, 10x looser bound
    # reductions here run over 131072+ rows per tensor: fp32 (atomic) summation noise, 2.5x the 32x32 bound for the
    # weights; a bias gradient is the plain sum of those rows (sign-like terms from the L1 loss cancel to a small
    # total, run-to-run 3e-4..6e-4 with the atomics' order): 5x
    worst = max(((k, v) for k, v in gerr.items() if k not in zero and not k.endswith(".b")), key=lambda kv: kv[1])
    assert worst[1] <= 2.5 * TOL_GRAD, worst
    worst_b = max(((k, v) for k, v in gerr.items() if k not in zero and k.endswith(".b")), key=lambda kv: kv[1])
    assert worst_b[1] <= 5 * TOL_GRAD, worst_b
    assert all(gerr[k] <= 2 * TOL_GRAD_ZERO for k in zero), {k: gerr[k] for k in zero if gerr[k] > 2 * TOL_GRAD_ZERO}
    x8 = eng.to_device(np.random.default_rng(2).uniform(0, 255, (8, 256, 256, 3)))
    vals = []
    for step in range(8):          # same seed every step: identical noise / dropout draws, so the losses are comparable
        eng.train_step(x8, 0.003, COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], 1.0, seed=7)
        m = eng.metrics()
        vals.append(1000.0 * m["r_exp"] + 10.0 * m["vae_kl_loss"])
    assert np.isfinite(vals).all() and min(vals[-3:]) < vals[0], vals


def test_golden_fixture_tiny():
    """The committed fixture (tests/golden/tiny_case.npz, generated by tests/golden/make_golden.py from the oracle)."""
    f = np.load(os.path.join(ROOT, "tests", "golden", "tiny_case.npz"))
    io = make_inputs("tiny", 4)
    assert np.array_equal(f["x"], io["x"])
    eng = _engine("tiny", 4)
    eng.set_params({k: f["p/" + k] for k in eng.param_table}); eng.set_state(io["state"])
    d = {k: eng.to_device(f[k]) for k in ("x", "eps", "noise", "keep")}
    out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "losses"))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    assert np.abs(out["recon"].cpu().numpy() - f["recon"]).max() <= TOL_RECON_ABS
    assert rel_err(out["losses"].cpu().numpy()[:, :3], f["losses"]) <= TOL_ELBO
    g = eng.get_grads()
    ref = {k: f["g/" + k] for k in eng.param_table}
    gerr = grad_errors(g, ref)
    zero = structurally_zero(ref)
    worst = max(gerr.items(), key=lambda kv: kv[1] / (TOL_GRAD_ZERO if kv[0] in zero else TOL_GRAD))
    assert worst[1] <= (TOL_GRAD_ZERO if worst[0] in zero else TOL_GRAD), worst


def test_device_rng_statistics():
    """Timed runs draw epsilon / noise / dropout with Philox on the device: check the distributions."""
    eng = _engine("c32nb", 256)
    x = eng.to_device(np.random.default_rng(0).uniform(0, 255, (256, 32, 32, 3)))
    out = eng.forward(x, True, seed=1234, outputs=("mu", "log_var", "z"))
    eng.torch.cuda.synchronize()
    eps = eng.tensor("eps", 256).cpu().numpy()
    assert abs(eps.mean()) < 3 * 0.01 / np.sqrt(eps.size) * 3 and abs(eps.std() / 0.01 - 1) < 0.03
    z = out["z"].cpu().numpy(); mu = out["mu"].cpu().numpy(); lv = out["log_var"].cpu().numpy()
    assert rel_err(z, mu + np.exp(lv) * eps) < 1e-5
    noise = eng.tensor("noise", 256).cpu().numpy()
    assert abs(noise.mean()) < 0.01 and abs(noise.std() - 1) < 0.01
    keep = eng.tensor("keep_mask", 256).cpu().numpy()
    assert set(np.unique(keep)) <= {0.0, 1.0} and abs(keep.mean() - 0.9) < 0.05
    out2 = eng.forward(x, True, seed=1235, outputs=("z",))
    assert not np.array_equal(out2["z"].cpu().numpy(), z)


def test_full_size_properties_c32nb_b512():
    """BASELINE config 2 (C32-nb, B=512) at full size: size-independent properties instead of the (slow) oracle --
    finite outputs, recon in range, per-sample independence of everything outside BatchNorm (batch-permutation
    equivariance of the loss vector), and the loss goes down under the optimiser."""
    from multiscale_variational_autoencoder_amd.initializers import init_params
    B = 512
    eng = _engine("c32nb", B)
    eng.set_params(init_params(eng.param_table, 42))
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 255, (B, 32, 32, 3)).astype(np.float32)
    eps = (rng.standard_normal((B, 48)) * 0.01).astype(np.float32)
    xd, ed = eng.to_device(x), eng.to_device(eps)
    o1 = eng.forward(xd, False, ed, outputs=("recon", "losses"))
    r1, l1 = o1["recon"].cpu().numpy(), o1["losses"].cpu().numpy()
    assert np.isfinite(r1).all() and r1.min() >= 0.0 and r1.max() <= 255.0 and np.isfinite(l1).all()
    perm = rng.permutation(B)
    o2 = eng.forward(eng.to_device(x[perm]), False, eng.to_device(eps[perm]), outputs=("losses",))
    assert rel_err(o2["losses"].cpu().numpy(), l1[perm]) < 1e-5
    first = last = None
    for step in range(12):
        eng.train_step(xd, 0.01, COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], 1.0, seed=step)
        m = eng.metrics()
        val = 1000.0 * m["r_exp"] + 10.0 * m["vae_kl_loss"]
        first = val if first is None else first
        last = val
    assert np.isfinite(last) and last < first, (first, last)
    assert np.isfinite(eng.params.cpu().numpy()).all()
