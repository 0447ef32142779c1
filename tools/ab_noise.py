"""Noise floor of the fused-vs-separate comparison (tests/test_fusion_ab_gpu.py): the same bf16 configuration run several
times with the fused launches ON, and once OFF; prints the median / worst per-tensor gradient difference of every pair."""
import os, subprocess, sys, tempfile, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_fusion_ab_gpu import WORKER
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tmp = tempfile.mkdtemp()
runs = {}
for tag, flag in (("on1", "1"), ("on2", "1"), ("on3", "1"), ("off1", "0"), ("off2", "0")):
    out = os.path.join(tmp, tag + ".npz")
    e = dict(os.environ); e.update({"MVAE_FUSE_PW_CHAIN": flag, "MVAE_FUSE_DW_CONV0": flag})
    subprocess.run([sys.executable, "-c", WORKER, ROOT, "c64nb", sys.argv[2] if len(sys.argv) > 2 else "4", dt, out], check=True, env=e)
    runs[tag] = np.load(out)
keys = [k for k in runs["on1"].files if k.startswith("g/")]
for a, b in itertools.combinations(runs, 2):
    A, Bn = runs[a], runs[b]
    rms = np.sqrt(sum(float((Bn[k].astype(np.float64) ** 2).sum()) for k in keys) / sum(Bn[k].size for k in keys))
    errs = [float(np.linalg.norm((A[k].astype(np.float64) - Bn[k]).ravel()) / max(np.linalg.norm(Bn[k].ravel()), 0.1 * rms * np.sqrt(Bn[k].size))) for k in keys]
    rd = np.abs(A["recon"].astype(np.float64) - Bn["recon"])
    print("%s vs %s: median %.2e  p90 %.2e  worst %.3f  recon max %.2f rms %.3f" % (a, b, np.median(errs), np.percentile(errs, 90), max(errs), rd.max(), np.sqrt((rd ** 2).mean())))
