"""GPU: the bfloat16-activation path (MVAE_ACT_BF16; BASELINE configs 4-5: 256x256x3, 7 scales, bf16) against the
float64 oracle.  Storage of the wide [M,c] tensors is bfloat16, products run on v_mfma_f32_32x32x16_bf16 with float32
accumulation; parameters, gradients, BatchNorm statistics and losses stay float32.

The bar ("bf16 configs: report error, expect ~1e-2"), stated here and asserted below:
  ELBO and its terms               <= 2e-2 relative
  reconstruction                   <= 1.5e-2 * 255 RMS, 0.15 * 255 max over pixels (measured 0.10: the tail of ~10^6 pixel-channels) (the decoder's BatchNorm subtracts the
                                   batch mean of a bf16-stored tensor: a relative rounding error of 2^-9 of the VALUE
                                   becomes (mean / std) times that after normalisation; measured 0.9 % RMS)
  forward tensors (saved t0/t1/out) <= 2e-2 relative (norm-wise)
  gradients, per tensor            <= 0.12 * max(||ref||, 0.1 * rms * sqrt(n)) for weight tensors (ndim >= 2) at batch 2
                                   (measured 0.09 worst, 0.03 at batch 32; the depthwise backward takes its ReLU mask
                                   from the LSB of dt2, which is therefore rounded to 7 significant bits: MVAE_LSB_MASK=0
                                   keeps 8 bits and reads t1 instead, +4 % step time), 0.5 for
                                   bias / BatchNorm vectors (column sums of bf16-rounded gradients with heavy
                                   cancellation: at batch 2 the sum of 8192 rounded values of mixed sign carries
                                   ~2^-9 * sqrt(sum v^2) of noise against a small net sum), median over all <= 1.5e-2
The oracle differentiates along the device's own branch at every kink (tests/common.py:device_kink_masks, now including
the L1 loss signs and the clip mask): with bf16 forward errors of ~3e-3 a visible share of the ReLU units and loss signs
sit on the other side than in float64, which is a property of the rounding, not of the kernels; the share is reported and
bounded separately.  The float32 path keeps its own, 100x tighter bars (tests/test_parity_gpu.py)."""
import json
import os

import numpy as np
import pytest

from tests.common import (COMPILE, CONFIGS, ROOT, device_kink_masks, engine_args, grad_errors, make_inputs,
                          oracle_config, reg_grad, rel_err, structurally_zero)

pytestmark = pytest.mark.gpu

TOL16_ELBO = 2e-2
TOL16_RECON_MAX = 0.15 * 255.0
TOL16_RECON_RMS = 1.5e-2 * 255.0
TOL16_FWD = 2e-2
TOL16_GRAD = 0.12
TOL16_GRAD_VEC = 0.5
TOL16_GRAD_MEDIAN = 1.5e-2
KINK16_MAX_FRACTION = 2e-2      # share of ReLU / hard-sigmoid units on the other side of the kink than in float64
KINK16_MAX_DISTANCE = 0.25      # and how far from the kink such a unit may be (activations are O(1))
OUT = os.path.join(ROOT, "gpurun_out")


def _engine(name, B):
    from multiscale_variational_autoencoder_amd.engine import Engine
    return Engine(**engine_args(name, B), act_dtype="bf16").bind()


def _compare(orc, eng, io, out, B, zero=None):
    """Device results against one oracle (float64-exact, or float64 with the device's bfloat16 roundings): the metric dict."""
    inter = {}
    res, G = orc.loss_and_grads(io["params"], io["state"], io["x"], io["eps"], io["noise"], io["keep"],
                                COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], inter=inter)
    rep = {"kink": orc.kink_report()}
    losses = out["losses"].cpu().numpy().astype(np.float64)
    elbo = (COMPILE["r_loss_factor"] * losses[:, 1] + COMPILE["kl_loss_factor"] * losses[:, 2]).mean()
    rep["elbo_rel"] = abs(elbo - res["data_loss"]) / abs(res["data_loss"])
    rep["r_rel"] = rel_err(losses[:, 0], res["r"]); rep["r_exp_rel"] = rel_err(losses[:, 1], res["r_exp"])
    rep["kl_rel"] = rel_err(losses[:, 2], res["kl"])
    rep["kl_scale_rel"] = rel_err(losses[:, 3:], res["kl_scale"])
    dr = out["recon"].cpu().numpy().astype(np.float64) - res["recon"]
    rep["recon_max"] = float(np.abs(dr).max()); rep["recon_rms"] = float(np.sqrt((dr ** 2).mean()))
    rep["mu_rel"] = rel_err(out["mu"].cpu().numpy(), res["mu"])
    fwd = {}
    for k, v in inter.items():
        if not k.endswith((".t0", ".t1", ".out", ".conv_base", ".conv", ".convT", ".dense")):
            continue
        ref = v.detach().numpy()
        if ref.ndim == 4:
            ref = np.transpose(ref, (0, 2, 3, 1))
        fwd[k] = rel_err(eng.tensor(k, B).cpu().numpy().reshape(ref.shape), ref)
    rep["fwd_worst"] = max(fwd.items(), key=lambda kv: kv[1])
    rep["fwd_all"] = fwd
    rep["fwd_median"] = float(np.median(list(fwd.values())))
    g = eng.get_grads()
    rg = reg_grad(io["params"], eng.param_table)
    gerr = grad_errors({k: g[k].astype(np.float64) + rg[k] for k in G}, G)
    zero = structurally_zero(G) if zero is None else zero     # (from the EXACT oracle: with the roundings modelled those
    rep["zero"] = sorted(zero)                                 #  gradients are rounding noise, not zero)
    real = {k: v for k, v in gerr.items() if k not in zero}
    vec = {k for k in real if len(eng.param_table[k]["shape"]) < 2}
    rep["grad_worst"] = sorted(((k, v) for k, v in real.items() if k not in vec), key=lambda kv: -kv[1])[:8]
    rep["grad_vec_worst"] = sorted(((k, v) for k, v in real.items() if k in vec), key=lambda kv: -kv[1])[:8]
    rep["grad_median"] = float(np.median(list(real.values())))
    rep["grad_p90"] = float(np.percentile(list(real.values()), 90))
    rep["grad_zero_worst"] = max(((k, gerr[k]) for k in zero), key=lambda kv: kv[1]) if zero else None
    return rep


def _device_tensors(eng, oc, B):
    """The device's stored forward tensors, by the names Oracle.set_forcing understands."""
    f = {}
    names = []
    for s in range(len(oc.z_dims)):
        names += ["enc%d.conv_base" % s, "dec%d.dense" % s, "enc%d.z" % s]
    for k in eng.param_table:
        if k.endswith(".mn.conv0.w"):
            p = k[:-len(".conv0.w")]
            names += [p + "." + t for t in ("t0", "t1", "out", "gap", "g")]
        elif k.endswith((".conv.w", ".convT.w")):
            names.append(k[:-2])
    for n in names:
        f[n] = eng.tensor(n, B).cpu().numpy().astype(np.float64)
    return f


def _report(name, B, expect_bf16_scales):
    """Two comparisons of one device run: against the float64 oracle (= the rounding error of the bf16 path, REPORTED and
    loosely bounded) and against the float64 oracle that rounds where the device rounds (Oracle(storage=...): what is left
    is the kernels' own error, tightly bounded -- rep["q"])."""
    from oracle.mvae_oracle import Oracle
    io = make_inputs(name, B)
    oc = oracle_config(name)
    eng = _engine(name, B)
    dts = eng.scale_dtypes()
    assert dts[:expect_bf16_scales] == ["bf16"] * expect_bf16_scales, dts
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    out = eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=("recon", "mu", "log_var", "losses"))
    eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
    eng.sync()
    masks = device_kink_masks(eng, B, io["x"], oc.min_value, oc.max_value)
    orc = Oracle(oc)
    orc.set_kink_masks(masks)
    rep = _compare(orc, eng, io, out, B)
    rep["scale_dtypes"] = dts
    orq = Oracle(oc, storage=dts, lsb_mask=os.environ.get("MVAE_LSB_MASK", "1") != "0")
    orq.set_kink_masks(masks)
    orq.set_forcing(_device_tensors(eng, oc, B))     # every stage continues from the device's stored tensors
    rep["q"] = _compare(orq, eng, io, out, B, zero=set(rep["zero"]))
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_bf16_%s_b%d.json" % (name, B)), "w") as f:
        json.dump(rep, f, indent=1, default=str)
    print(json.dumps(rep, default=str))
    return rep, eng


# bars against the rounding-aware oracle: what the KERNELS add on top of the storage roundings
TOLQ_FWD = 2e-3
TOLQ_GRAD = 1e-2
TOLQ_GRAD_VEC = 5e-2
KINKQ_MAX_FRACTION = 1e-4


def _check_q(rep):
    q = rep["q"]
    kr = q["kink"]
    assert kr["flips"] <= KINKQ_MAX_FRACTION * kr["units"], kr
    for k in ("elbo_rel", "r_rel", "r_exp_rel", "kl_rel", "kl_scale_rel"):
        assert q[k] <= 2e-3, (k, q[k])
    assert q["fwd_worst"][1] <= TOLQ_FWD, q["fwd_worst"]
    assert q["grad_worst"][0][1] <= TOLQ_GRAD, q["grad_worst"]
    assert q["grad_vec_worst"][0][1] <= TOLQ_GRAD_VEC, q["grad_vec_worst"]


def _check(rep, tol_grad=TOL16_GRAD, tol_vec=TOL16_GRAD_VEC):
    kr = rep["kink"]
    assert kr["flips"] <= KINK16_MAX_FRACTION * kr["units"], kr
    assert kr["max_abs_at_flip"] <= KINK16_MAX_DISTANCE, kr
    for k in ("elbo_rel", "r_rel", "r_exp_rel", "kl_rel", "kl_scale_rel"):
        assert rep[k] <= TOL16_ELBO, (k, rep[k])
    assert rep["recon_max"] <= TOL16_RECON_MAX and rep["recon_rms"] <= TOL16_RECON_RMS, (rep["recon_max"], rep["recon_rms"])
    assert rep["fwd_worst"][1] <= TOL16_FWD, rep["fwd_worst"]
    assert rep["grad_worst"][0][1] <= tol_grad, rep["grad_worst"]
    assert tol_vec is None or rep["grad_vec_worst"][0][1] <= tol_vec, rep["grad_vec_worst"]
    assert rep["grad_p90"] <= 3e-2, rep["grad_p90"]
    assert rep["grad_median"] <= TOL16_GRAD_MEDIAN, rep["grad_median"]


# ("c64nb", 32): the scale-0 decoder then has 131072 rows per tensor, as many as C256-nb at batch 2 below
@pytest.mark.timeout(900)
@pytest.mark.parametrize("name,B,nbf", [("c32nb", 8, 2), ("c64nb", 2, 3), ("c64nb", 32, 3), ("c96nb", 8, 2)])
def test_bf16_forward_backward_parity(name, B, nbf):
    rep, eng = _report(name, B, nbf)
    if B == 2:
        # two images: the decoder's bias gradients (and their leak into the 1x1 weight gradients) are column sums of a
        # nearly zero-mean field of bf16-rounded terms -- a noise statistic.  The forward pass has float atomics (GAP,
        # BatchNorm sums), a last-bit difference there re-rolls bf16 roundings downstream: the same build measured
        # 0.39 and 0.59 (vectors), 0.090 and 0.115 (weights) in two runs.  Batch 32 (next case) holds the standard bars.
        _check(rep, tol_grad=0.2, tol_vec=1.0)
    else:
        _check(rep)
    _check_q(rep)              # the kernels' own error: the same bars at every batch


@pytest.mark.timeout(1500)
def test_bf16_c256nb_parity_and_full_batch_properties():
    """BASELINE config 4's model in bf16: batch 2 against the oracle, then the full batch 64: finite, reconstruction in
    range, the loss goes down under the optimiser (size-independent properties: the oracle is too slow at B = 64)."""
    name = "c256nb"
    rep, eng = _report(name, 2, 5)              # 256 .. 16 wide scales in bf16, the 8x8 and 4x4 tops in float32
    # Two images of 256x256: the gradient field that reaches the scale-0 decoder is (by the BatchNorm backward) zero-mean
    # over the pixels of a channel, so its column sums (bias gradients) and their leak into the 1x1 weight gradients
    # through the mean activation are sums of 131072 bf16-rounded terms that cancel almost completely: measured 0.26
    # (weights) and 2.1 (bias vectors) relative on dec0, 7e-4 median.  The same number of rows spread over 32 images
    # (c64nb, B = 32 above) gives 0.03 / 0.10 with the same kernels: the bar here is the batch-2 statistics, not the
    # arithmetic, so the vectors are reported and only the weights, the median and the 90th percentile are bounded.
    _check(rep, tol_grad=0.35, tol_vec=None)
    _check_q(rep)
    eng.close()
    B = 64
    from multiscale_variational_autoencoder_amd.engine import Engine
    from multiscale_variational_autoencoder_amd.initializers import init_params
    x_host = np.random.default_rng(2).uniform(0, 255, (B, 256, 256, 3))
    traj = {}
    for dt in ("f32", "bf16"):     # same init, batch and seeds: the bf16 run has to follow the float32 run step by step
        if dt == "bf16":
            eng = _engine(name, B)
        else:
            eng = Engine(**engine_args(name, B)).bind()
        eng.set_params(init_params(eng.param_table, 42))
        x = eng.to_device(x_host)
        vals = []
        for step in range(8):      # same seed every step: identical noise / dropout draws, so the losses are comparable
            eng.train_step(x, 0.003, COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"], 1.0, seed=7)
            m = eng.metrics()
            vals.append(1000.0 * m["r_exp"] + 10.0 * m["vae_kl_loss"])
        traj[dt] = np.array(vals)
        if dt == "f32":
            eng.close()
    # Adagrad's first update moves every weight by lr, so on this noise batch the objective jumps at step 1 and then
    # descends with an oscillation, in float32 as well (tools/bf16_train_ab.py): the property is "tracks float32 within
    # 3 % at every step and comes down from the jump", not monotone descent from step 0.
    assert np.isfinite(traj["bf16"]).all(), traj
    assert np.abs(traj["bf16"] / traj["f32"] - 1.0).max() <= 0.03, traj
    assert traj["bf16"][-2:].min() < traj["bf16"][1:4].max(), traj
    out = eng.forward(x, False, seed=3, outputs=("recon",))
    r = out["recon"].cpu().numpy()
    assert np.isfinite(r).all() and r.min() >= 0.0 and r.max() <= 255.0
    assert np.isfinite(eng.params.cpu().numpy()).all()
