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
                   A=176231857, P=36045205, F=13444.5e6),
}
HBM_PEAK = 8.0e12        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy ceiling)
FP32_PEAK = 157.3e12     # fp32 vector = f32-input MFMA rate


def algorithmic_bytes_per_step(w, B):
    """B*5*A*4 + 7*P*4 (fp32 activations; fwd write+read, bwd re-read, grad write+read;
    weight read x2, dW write+read, accumulator r/w, weight write)."""
    return B * 5 * w["A"] * 4 + 7 * w["P"] * 4


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
    cores = max(1, min(cores, 16))     # a 1-GPU box grants a 16-core share; more threads only oversubscribe it
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
    return dict(value=Bc * n / dt, unit="images/sec", cores=cores, kind="port",
                sample="%d train steps of batch %d (%s, fp32, torch-CPU restatement oracle/mvae_oracle.py, %d threads)"
                       % (n, Bc, wname, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c32nb", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    # rehearsal of the N > 1 path on a one-GPU box (MVAE_BENCH_REHEARSE=gloo): every rank uses cuda:0 and the
    # all-reduce goes through gloo -- exercises rank handling, the all-reduce of the reduce arena and the JSON line,
    # not RCCL performance.  The driver's real runs use backend "nccl" (= RCCL), one rank per GPU.
    rehearse = os.environ.get("MVAE_BENCH_REHEARSE", "")
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(rehearse, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from multiscale_variational_autoencoder_amd.engine import Engine
    from multiscale_variational_autoencoder_amd.initializers import init_params
    from multiscale_variational_autoencoder_amd import _abi
    w = WORKLOADS[args.workload]
    B = args.batch or w["batch"]
    eng = Engine(w["input_dims"], w["z_dims"], w["encoder"], w["decoder"], 0.0, 255.0, 0.01, B).bind(local)
    eng.set_params(init_params(eng.param_table, 42))          # identical replicas on every rank
    H, Wd, C = w["input_dims"]
    x = eng.to_device(np.random.default_rng(1234 + rank).uniform(0, 255, (B, H, Wd, C)))
    lr, rf, kf, clip = 1e-3, 1000.0, 10.0, 1.0

    def step(i):
        eng.train_step(x, lr, rf, kf, clip, seed=1000 + i)

    for i in range(max(args.warmup, 1)):     # the first call of each signature captures its hipGraph
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    m = eng.metrics()
    finite = bool(np.isfinite(m["r_exp"]) and np.isfinite(m["vae_kl_loss"]))
    # "also report forward+backward+ELBO without the optimiser" (secondary number, single GPU only)
    fb_ms = None
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
        lib = _abi.load_library()
        lib.mvae_profile_enable(1)
        nprof = 3
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
        dom = max(kernels, key=lambda k: kernels[k]["ms"])
        d = kernels[dom]
        bytes_per_launch = d["bytes"] / d["count"]
        dur = d["ms"] * 1e-3 / d["count"]
        # HBM bytes per launch from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE runs of this
        # same command, FETCH doubled per MI355X_MICROARCH.md; tools/pmc_traffic.py), when that kernel was profiled
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic = pmc.get("mvae::" + dom, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            pass
        roofline = dict(kernel=dom, bound="hbm", achieved=bytes_per_launch / dur / 1e9, peak=HBM_PEAK / 1e9,
                        unit="GB/s", frac=bytes_per_launch / dur / HBM_PEAK, traffic=traffic,
                        avg_launch_us=dur * 1e6, launches_per_step=d["count"] / nprof,
                        algorithmic_bytes_per_launch=bytes_per_launch,
                        flop_frac=d["flops"] / d["count"] / dur / FP32_PEAK, share_of_step=d["share"])

    ms = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    step_bytes = algorithmic_bytes_per_step(w, B)
    out = {
        "metric": "images/sec (train step, fwd+bwd+ELBO)", "value": value, "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %dx%dx%d, %d scales, z=%s, enc/dec filters %s, batch %d per GPU (global %d), "
                               "train step incl. Adagrad" % (args.workload, H, Wd, C, len(w["z_dims"]), w["z_dims"][0],
                                                            w["encoder"]["filters"], B, B * world),
                   "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                   "collective": "1 RCCL all-reduce of %d floats per step" % eng.R if world > 1 else "none"},
        "finite": finite,
        "fwd_bwd_only": None if fb_ms is None else {"ms_per_step": fb_ms, "images_per_sec": B / (fb_ms * 1e-3)},
        "elbo_metrics": {k: float(v) for k, v in m.items()},
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "hbm_frac": step_bytes / (dt / args.steps) / HBM_PEAK,
                          "fp32_flop_frac": 3 * w["F"] * B / (dt / args.steps) / FP32_PEAK},
        "roofline": roofline,
    }
    if rank == 0:
        if kernels is not None:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_kernels_%s_n%d.json" % (args.workload, world)), "w") as f:
                json.dump(kernels, f, indent=1, sort_keys=True)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
