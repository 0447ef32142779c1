#!/usr/bin/env python3
"""
bench.py -- images/sec of the multiscale-VAE TRAIN STEP (forward + ELBO + backward + [RCCL all-reduce] +
per-variable clipnorm + Adagrad) on MI355X, the metric BASELINE.json names.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

Workload (BASELINE.json configs[1], weak-scaled per configs[2]): C32-nb = input 32x32x3, 3 scales, z=16/scale,
notebook encoder/decoder dicts, fp32, batch 512 PER GPU, synthetic U[0,255) images, glorot weights, on-device
Philox noise / dropout / epsilon, compile(lr=1e-3, r=1000, kl=10, clipnorm=1).  Inputs are resident in HBM before
the timed region.  One JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NB = {"filters": [64, 64, 64, 64, 32],
      "kernel_size": [(5, 5), (3, 3), (3, 3), (1, 1), (1, 1)],
      "strides": [(2, 2), (1, 1), (1, 1), (1, 1), (1, 1)]}
WORKLOADS = {
    "c32nb": dict(input_dims=(32, 32, 3), z_dims=[16, 16, 16], encoder=NB, decoder=NB, batch=512,
                  # A activation elems / image, trainable params, forward FLOPs / image
                  A=2716945, P=1295625, F=207.2e6),
    "c32def": dict(input_dims=(32, 32, 3), z_dims=[16, 16, 16],
                   encoder={"filters": [32], "kernel_size": [(3, 3)], "strides": [(1, 1)]},
                   decoder={"filters": [32], "kernel_size": [(3, 3)], "strides": [(1, 1)]}, batch=512,
                   A=592657, P=2138313, F=19.9e6),
    "c256nb": dict(input_dims=(256, 256, 3), z_dims=[16] * 7, encoder=NB, decoder=NB, batch=64,
                   A=176231857, P=36045205, F=13444.5e6, dtype="bf16"),
}
HBM_PEAK = 8.0e12        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
FP32_PEAK = 157.3e12     # fp32 vector = f32-input MFMA rate
BF16_PEAK = 2.5e15       # dense bf16 MFMA


def algorithmic_bytes_per_step(w, B, act="f32"):
    """B*5*A*dt + 7*P*4 (dt = 4 fp32 / 2 bf16 activations; fwd write+read, bwd re-read, grad
    write+read; weight read x2, dW write+read, accumulator r/w, weight write -- parameters stay fp32)."""
    return B * 5 * w["A"] * (2 if act == "bf16" else 4) + 7 * w["P"] * 4


def cpu_baseline(wname, seconds=20.0):
    """The oracle ('port' kind: torch-CPU fp32 restatement of the identical train step) timed on this box's host
    cores on a bounded sample: batches of 128 images of the same workload until ~`seconds` of CPU work."""
    import torch
    from oracle.mvae_oracle import Oracle, OracleConfig, param_table
    from multiscale_variational_autoencoder_amd.initializers import init_params, init_state
    from collections import OrderedDict
    w = WORKLOADS[wname]
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a 1-GPU box grants a 16-core share of a much larger host: the affinity mask shows every host core, the cgroup
    # quota (when readable) the real share -- more threads than that only thrash (uncapped, one step took minutes)
    quota = 16
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, quota, 64))
    torch.set_num_threads(cores)
    oc = OracleConfig(w["input_dims"], w["z_dims"], encoder=w["encoder"], decoder=w["decoder"])
    P, S = param_table(oc)
    p = init_params(OrderedDict((k, dict(shape=v[0])) for k, v in P.items()), 42)
    st = init_state(OrderedDict((k, dict(shape=v)) for k, v in S.items()))
    a = {k: np.full(v.shape, 0.1, np.float32) for k, v in p.items()}
    Bc = 128 if wname != "c256nb" else 4
    rng = np.random.default_rng(1234)
    H, Wd, C = w["input_dims"]
    x = rng.uniform(0, 255, (Bc, H, Wd, C)).astype(np.float32)
    eps = (rng.standard_normal((Bc, sum(w["z_dims"]))) * 0.01).astype(np.float32)
    noise = rng.standard_normal(x.shape).astype(np.float32)
    keep = (rng.uniform(size=(Bc, C)) >= 0.1).astype(np.float32)
    orc = Oracle(oc, dtype=torch.float32)
    step = lambda: orc.train_step(p, a, st, x, eps, noise, keep, 1e-3, 1000.0, 10.0, 1.0)
    step()                                           # warm-up (allocator, oneDNN primitives)
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds and n < 50:
        step(); n += 1
        sys.stderr.write("cpu_baseline step %d  %.1fs\n" % (n, time.time() - t0)); sys.stderr.flush()
    dt = time.time() - t0
    return dict(value=Bc * n / dt, unit="images/sec", cores=cores, kind="port", cpu_model=cpu_model(),
                sample="%d train steps of batch %d (%s, fp32, torch-CPU restatement oracle/mvae_oracle.py -- not Keras --, "
                       "%d threads on %s)" % (n, Bc, wname, cores, cpu_model()))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_steps(eng, step, steps, torch, dist, world):
    """EXACTLY `steps` steps between barrier + synchronize on both sides (the driver's contract: wall clock, max over
    ranks) with a HIP event recorded on the engine's stream after every step (hipEvent timing, median)."""
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs[0].record(eng.stream)
    for i in range(steps):
        step(i)
        evs[i + 1].record(eng.stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(steps)])
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, per


def self_launch(n):
    """`python bench.py --gpus N` without an external launcher: the parent -- which never touches the GPU (no torch
    import, no HIP call) -- starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a
    child process, relays its output (rank 0 prints the one JSON line) and returns its exit code (torchrun's is non-zero
    as soon as any rank's is)."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: self-launch: %s\n" % " ".join(cmd))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def kernel_profile(step, nprof, act, workload):
    """`nprof` instrumented steps (HIP events around every launch, on the launch stream, scales one after another):
    ({tag: {count, ms, bytes, flops, avg_us, share, GBps, TFLOPs}}, roofline block of the dominant kernel)."""
    from multiscale_variational_autoencoder_amd import _abi
    lib = _abi.load_library()
    lib.mvae_profile_enable(1)
    for i in range(nprof):
        step(10_000 + i)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.mvae_profile_report(buf, len(buf))
    lib.mvae_profile_enable(0)
    kernels = json.loads(buf.value.decode())
    tot = sum(v["ms"] for v in kernels.values())
    for v in kernels.values():
        v["avg_us"] = 1e3 * v["ms"] / v["count"]
        v["share"] = v["ms"] / tot
        v["GBps"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9
        v["TFLOPs"] = v["flops"] / (v["ms"] * 1e-3) / 1e12
    pmc = pmc_traffic_table(workload)
    fam_alg, fam_n = {}, {}
    for k, v in kernels.items():
        fam_alg[k.split("@")[0]] = fam_alg.get(k.split("@")[0], 0.0) + v["bytes"]
        fam_n[k.split("@")[0]] = fam_n.get(k.split("@")[0], 0) + v["count"]
    for k, v in kernels.items():       # measured HBM bytes per launch (committed PMC passes) next to the algorithmic ones
        fam = k.split("@")[0]          # size-bucketed tags (conv1x1_f@2^17 ...) share one rocprof kernel family: the
        t = pmc_lookup(pmc[0], fam) if pmc else None       # PMC figure is then the family's launch-weighted mean
        v["algorithmic_bytes_per_launch"] = v["bytes"] / v["count"]
        v["hbm_bytes_per_launch_pmc"] = t if fam == k else None
        v["traffic_ratio"] = None if t is None else t / max(fam_alg[fam] / fam_n[fam], 1.0)
    dom = max(kernels, key=lambda k: kernels[k]["ms"])
    d = kernels[dom]
    bytes_per_launch = d["bytes"] / d["count"]
    dur = d["ms"] * 1e-3 / d["count"]
    peak_f = BF16_PEAK if act == "bf16" else FP32_PEAK
    roofline = dict(kernel=dom, bound="hbm", achieved=bytes_per_launch / dur / 1e9, peak=HBM_PEAK / 1e9,
                    unit="GB/s", frac=bytes_per_launch / dur / HBM_PEAK, traffic=d["hbm_bytes_per_launch_pmc"],
                    traffic_source=None if not pmc or d["hbm_bytes_per_launch_pmc"] is None else
                    "profiles/%s (separate rocprofv3 --pmc passes of this command, not this run)" % pmc[1],
                    avg_launch_us=dur * 1e6, launches_per_step=d["count"] / nprof,
                    algorithmic_bytes_per_launch=bytes_per_launch,
                    flop_frac=d["flops"] / d["count"] / dur / peak_f, share_of_step=d["share"])
    return kernels, roofline


def pmc_traffic_table(workload):
    """The newest committed rocprofv3 PMC summary of this workload (tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in
    separate passes, FETCH doubled per MI355X_MICROARCH.md), or None."""
    for cand in ("round4_pmc_traffic_%s.json" % workload, "round3_pmc_traffic_%s.json" % workload, "round2_pmc_traffic_%s.json" % workload):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                return json.load(f), cand
        except (OSError, ValueError):
            continue
    return None


# profiler tag -> rocprofv3 kernel-name prefixes it covers (a tag may cover several template instances)
TAG_KERNELS = {"conv1x1_f": ["k_gemm_rows"], "conv1x1_t": ["k_gemm_rows"], "conv1x1_wgrad": ["k_wgrad_rows"],
               "convkxk_f": ["k_conv_taps"], "convkxk_t": ["k_conv_taps"], "convkxk_wgrad": ["k_wgrad_taprow"],
               "se_fwd": ["k_se_fwd1", "k_se_fwd2"], "se_bwd": ["k_se_bwd1", "k_se_bwd2"], "k_conv2_tile": ["k_conv0_tile<64, true>", "k_conv0_tile<32, true>"],
               "k_conv0_tile": ["k_conv0_tile<64, false>", "k_conv0_tile<32, false>"], "k_conv2_chain": ["k_conv2_chain<false"],
               "k_conv2_chain3": ["k_conv2_chain<true"], "head_fwd": ["k_head_fwd"], "head_bwd": ["k_head_bwd_reduce", "k_head_bwd_apply"],
               "convbase_fwd": ["k_convbase_fwd_mfma"], "convbase_wgrad": ["k_convbase_wgrad_mfma"], "k_dw_fwd_img": ["k_dw_fwd_img"],
               "k_dw_bwd_img": ["k_dw_bwd_img"], "k_mn_fwd_img": ["k_mn_fwd_img"], "k_mn_bwd_img": ["k_mn_bwd_img"]}


def pmc_lookup(pmc, tag):
    """HBM bytes per launch of profiler tag `tag`: launch-weighted mean over the kernels the tag covers."""
    names = TAG_KERNELS.get(tag, [tag])
    fam = [v for k, v in pmc.items() for n in names if k.replace("mvae::", "").startswith(n)]
    if not fam and "<" not in tag:
        fam = [v for k, v in pmc.items() if k.replace("mvae::", "").startswith(tag + "<")]
    if not fam:
        return None
    return sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / sum(v["launches"] for v in fam)


def secondary_workload(wname, local, torch, dist, lr, rf, kf, clip, profile=True, steps=10):
    """One more workload of BASELINE.json beside the headline: {ms_per_step, images_per_sec, hbm_frac, roofline}."""
    from multiscale_variational_autoencoder_amd.engine import Engine
    from multiscale_variational_autoencoder_amd.initializers import init_params
    w = WORKLOADS[wname]
    B, act = w["batch"], w.get("dtype", "f32")
    H, Wd, C = w["input_dims"]
    e = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype=act).bind(local)
    e.set_params(init_params(e.param_table, 42))
    # (synthetic batch, resident in the handle's own input buffer: Engine.stage_input -- a training loop gathers into it)
    x = e.stage_input(e.to_device(np.random.default_rng(77).uniform(0, 255, (B, H, Wd, C))))
    st = lambda i: e.train_step(x, lr, rf, kf, clip, seed=9000 + i)
    for i in range(3):
        st(i)
    dt, per = timed_steps(e, lambda i: st(100 + i), steps, torch, dist, 1)
    nbytes = algorithmic_bytes_per_step(w, B, act)
    out = {"workload": "%s: %dx%dx%d, %d scales, batch %d, %s activations" % (wname, H, Wd, C, len(w["z_dims"]), B, act),
           "images_per_sec": B * steps / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "ms_per_step_median_events": float(np.median(per)), "algorithmic_bytes_per_step": nbytes,
           "hbm_frac": nbytes / (dt / steps) / HBM_PEAK, "scale_dtypes": e.scale_dtypes(),
           "flop_frac": 3 * w["F"] * B / (dt / steps) / (BF16_PEAK if act == "bf16" else FP32_PEAK)}
    m = e.metrics()
    out["finite"] = bool(np.isfinite(m["r_exp"]) and np.isfinite(m["vae_kl_loss"]))
    if profile:
        kernels, out["roofline"] = kernel_profile(st, 2, act, wname)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_%s_%s_n1.json" % (wname, act)), "w") as f:
            json.dump(kernels, f, indent=1, sort_keys=True)
    e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 50 timed steps unless the caller says otherwise
    # 100 untimed steps (0.5 s): the first process on a fresh box runs its first few hundred milliseconds 1.5 % slower
    # (tools/cold_warm.sh: 5.14 / 5.16 ms for the first two 60-step processes, 5.06 - 5.08 afterwards and with --warmup 300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c32nb", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--dtype", default="", choices=["", "f32", "bf16"], help="activation storage (default: the workload's)")
    ap.add_argument("--force-collective", action="store_true",
                    help="one-GPU run through the RCCL all-reduce branch (a world-size-1 nccl group)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    # debugging switches that change what a step computes must not be set for a measurement (the release library does not
    # even compile MVAE_DEBUG_ONLY_SCALE in; MVAE_DEBUG_BUILD=1 at build time brings it back for tools/chain_only.py)
    debug_env = sorted(k for k in os.environ if k.startswith("MVAE_DEBUG_"))
    if debug_env:
        raise SystemExit("bench.py refuses to run with debugging switches set: %s" % ", ".join(debug_env))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # rehearsal of the N > 1 path on a one-GPU box (MVAE_BENCH_REHEARSE=gloo): every rank uses cuda:0 and the
    # all-reduce goes through gloo -- exercises rank handling, the all-reduce of the reduce arena and the JSON line,
    # not RCCL performance.  The driver's real runs use backend "nccl" (= RCCL), one rank per GPU.
    rehearse = os.environ.get("MVAE_BENCH_REHEARSE", "")
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if rehearse:
            dist.init_process_group(rehearse, rank=rank, world_size=world)
        else:          # any RCCL error raises and ends the process with a non-zero exit code
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from multiscale_variational_autoencoder_amd.engine import Engine
    from multiscale_variational_autoencoder_amd.initializers import init_params
    from multiscale_variational_autoencoder_amd import _abi
    if _abi.load_library().mvae_debug_build() != 0:
        raise SystemExit("bench.py refuses to measure a MVAE_DEBUG_BUILD=1 library: rebuild without it")
    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    act = args.dtype or w.get("dtype", "f32")
    eng = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype=act).bind(local)
    eng.set_params(init_params(eng.param_table, 42))          # identical replicas on every rank
    H, Wd, C = w["input_dims"]
    # the synthetic batch lives in the handle's own input buffer (Engine.stage_input): a training loop gathers every batch
    # straight into it (Engine.gather_batch), so no step pays a device-to-device copy of its input
    x = eng.stage_input(eng.to_device(np.random.default_rng(1234 + rank).uniform(0, 255, (B, H, Wd, C))))
    lr, rf, kf, clip = 1e-3, 1000.0, 10.0, 1.0
    coll_timing = []
    collective = eng.collective_active(args.force_collective)

    def step(i):        # rank-dependent device-RNG seed: replicas draw independent noise / dropout / epsilon
        eng.train_step(x, lr, rf, kf, clip, seed=(1000 + i) ^ (rank * 0x9E3779B97F4A7C15 & (2 ** 62 - 1)),
                       force_collective=args.force_collective, timing=coll_timing if collective else None)

    for i in range(max(args.warmup, 1)):     # the first call of each signature captures its hipGraph
        step(i)
    del coll_timing[:]
    dt, per_step = timed_steps(eng, lambda i: step(args.warmup + i), args.steps, torch, dist, world)
    coll = None
    if collective and coll_timing:
        ar_ms = float(np.median([a.elapsed_time(b) for a, b in coll_timing]))
        overlap = eng.dp_overlap_active()
        # overlap: the timed collective is the early message (Dense-weight gradients, on the comm stream, concurrent
        # with backward phase 1); the second message (the rest of the arena) follows on the engine's stream
        nbytes = (eng.reduce_split if overlap else eng.R) * 4
        n = max(world, 1)
        coll = {"collective_bytes": nbytes, "overlap_with_backward": overlap,
                "second_message_bytes": (eng.R - eng.reduce_split) * 4 if overlap else 0, "allreduce_ms_median": ar_ms,
                "algbw_GBps": nbytes / (ar_ms * 1e-3) / 1e9,
                "busbw_GBps": nbytes / (ar_ms * 1e-3) / 1e9 * (2.0 * (n - 1) / n), "backend": rehearse or "nccl(RCCL)",
                "world": world}
        if world == 1:
            coll["note"] = "world 1 (--force-collective): the all-reduce is an in-place no-op, the figure is its launch cost"
    m = eng.metrics()
    finite = bool(np.isfinite(m["r_exp"]) and np.isfinite(m["vae_kl_loss"]))
    coll_desc = ("none" if not collective else
                 "2 RCCL all-reduces per step: %d floats (Dense-weight gradients) overlapped with the encoder "
                 "half of the backward pass, then %d floats" % (eng.reduce_split, eng.R - eng.reduce_split)
                 if eng.dp_overlap_active() else "1 RCCL all-reduce of %d floats per step" % eng.R)
    # "also report forward+backward+ELBO without the optimiser" (secondary number, single GPU only)
    fb_ms = None
    secondary = {}
    if world == 1:
        nfb = max(min(args.steps, 10), 1)
        for i in range(2):
            eng.forward(x, True, seed=5000 + i, outputs=()); eng.backward(rf, kf)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nfb):
            eng.forward(x, True, seed=6000 + i, outputs=()); eng.backward(rf, kf)
        torch.cuda.synchronize()
        fb_ms = 1e3 * (time.perf_counter() - t1) / nfb

    # ---- per-kernel durations, live, with HIP events on the launch stream (instrumented extra steps)
    roofline, kernels = None, None
    if not args.no_kernel_profile:
        kernels, roofline = kernel_profile(step, 3, act, args.workload)

    # BASELINE config 1 (C32-nb at batch 128, the reference's own notebook batch) on the GPU, next to the headline
    if world == 1 and args.workload == "c32nb" and not args.no_secondary and not args.batch:
        e2 = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, 128, act_dtype=act).bind(local)
        e2.set_params(init_params(e2.param_table, 42))
        x2 = e2.stage_input(e2.to_device(np.random.default_rng(99).uniform(0, 255, (128, H, Wd, C))))
        s2 = lambda i: e2.train_step(x2, lr, rf, kf, clip, seed=7000 + i)
        for i in range(5):
            s2(i)
        dt2, per2 = timed_steps(e2, lambda i: s2(100 + i), 30, torch, dist, 1)
        secondary["c32nb_b128"] = {"images_per_sec": 128 * 30 / dt2, "ms_per_step": 1e3 * dt2 / 30,
                                   "ms_per_step_median_events": float(np.median(per2)),
                                   "hbm_frac": algorithmic_bytes_per_step(w, 128, act) / (dt2 / 30) / HBM_PEAK}
        e2.close()
        if act == "f32":       # the same workload and batch with bfloat16 activation storage (not the headline: BASELINE's
            # config 2 is float32) -- its own algorithmic bytes (2 bytes per activation element)
            e3 = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B, act_dtype="bf16").bind(local)
            e3.set_params(init_params(e3.param_table, 42))
            x3 = e3.stage_input(x)
            s3 = lambda i: e3.train_step(x3, lr, rf, kf, clip, seed=8000 + i)
            for i in range(5):
                s3(i)
            dt3, per3 = timed_steps(e3, lambda i: s3(100 + i), 30, torch, dist, 1)
            secondary["c32nb_b%d_bf16" % B] = {"images_per_sec": B * 30 / dt3, "ms_per_step": 1e3 * dt3 / 30,
                                               "ms_per_step_median_events": float(np.median(per3)),
                                               "hbm_frac": algorithmic_bytes_per_step(w, B, "bf16") / (dt3 / 30) / HBM_PEAK}
            e3.close()

    # BASELINE config 4 (C256-nb, 7 scales, batch 64, bf16 activations) under the driver's default run: its own step time,
    # roofline fraction and dominant-kernel block (about 2 s of GPU time; the headline engine is released first)
    if world == 1 and args.workload == "c32nb" and not args.no_secondary and not args.batch and act == "f32":
        eng.close()
        del eng, x
        torch.cuda.empty_cache()
        try:
            secondary["c256nb_b64_bf16"] = secondary_workload("c256nb", local, torch, dist, lr, rf, kf, clip,
                                                              profile=not args.no_kernel_profile)
        except Exception as e:        # never lose the headline line to the secondary one
            secondary["c256nb_b64_bf16"] = {"error": "%s: %s" % (type(e).__name__, e)}

    ms = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    step_bytes = algorithmic_bytes_per_step(w, B, act)
    med = float(np.median(per_step))
    out = {
        "metric": "images/sec (train step: fwd+bwd+ELBO+clipnorm-Adagrad)", "value": value, "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": act, "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d, %d scales, z=%s, enc/dec filters %s, batch %d per GPU (global %d), "
                               "train step incl. Adagrad, %s activations" % (args.workload, H, Wd, C, len(w["z_dims"]), w["z_dims"][0],
                                                            w["encoder"]["filters"], B, B * world, act),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                   "collective": coll_desc},
        "finite": finite, "debug_env": debug_env,
        # float32 5x5 convolutions as split-bf16 products (csrc/kernels_split.hip): status 1 = in use, 0 = switched off, 2 =
        # disabled by the bind-time hardware self-test; erratum = wrong values the self-test's v_pk_fma_f32 check kernel
        # returned on this board (the build emits no packed-float32 instructions because of it)
        "split_conv": {"status": int(_abi.load_library().mvae_split_conv_status()),
                       "packed_f32_erratum_wrong_values": int(_abi.load_library().mvae_split_conv_erratum())},
        "timing": {"ms_per_step_median_events": med, "ms_per_step_min_events": float(per_step.min()),
                   "ms_per_step_p90_events": float(np.percentile(per_step, 90)),
                   "images_per_sec_median_events": B * world / (med * 1e-3),
                   "note": "value / ms_per_step: wall clock over the K steps between barrier+synchronize, max over ranks "
                           "(driver contract); *_events: HIP events on the engine's stream after every step (rank 0)"},
        "fwd_bwd_only": None if fb_ms is None else {"ms_per_step": fb_ms, "images_per_sec": B / (fb_ms * 1e-3)},
        "elbo_metrics": {k: float(v) for k, v in m.items()},
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "hbm_frac": step_bytes / (dt / args.steps) / HBM_PEAK,
                          "hbm_frac_median_events": step_bytes / (med * 1e-3) / HBM_PEAK,
                          "flop_frac": 3 * w["F"] * B / (dt / args.steps) / (BF16_PEAK if act == "bf16" else FP32_PEAK),
                          "flop_peak": "bf16 MFMA 2.5 PF" if act == "bf16" else "fp32 157.3 TF"},
        "roofline": roofline,
        "collective": coll,
        "secondary": secondary or None,
    }
    if rank == 0:
        if kernels is not None:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_%s_%s_n%d.json" % (args.workload, act, world)), "w") as f:
                json.dump(kernels, f, indent=1, sort_keys=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
