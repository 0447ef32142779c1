"""Fused vs separate launches at several batches of c32nb float32 (the A/B of tests/test_fusion_ab_gpu.py, printed instead of
asserted): is the difference at batch 300 -- blocks of the fused kernels walking two images -- any different from batch 256,
where every block walks one?   python tools/fusion_ab_batch.py 256 300 512"""
import os, sys, pathlib, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from tests.test_fusion_ab_gpu import _run
ON = {"MVAE_FUSE_PW_CHAIN": "1", "MVAE_FUSE_DW_CONV0_F32": "1", "MVAE_SPLIT_DUAL": "1", "MVAE_FUSE_MN_FWD": "1"}
OFF = {"MVAE_FUSE_PW_CHAIN": "0", "MVAE_FUSE_DW_CONV0_F32": "0", "MVAE_SPLIT_DUAL": "0", "MVAE_FUSE_MN_FWD": "0"}
for B in [int(a) for a in sys.argv[1:]] or [256, 300]:
    tmp = pathlib.Path(tempfile.mkdtemp())
    runs = {"on": _run(tmp, "on", ON, "c32nb", B, "f32"), "off": _run(tmp, "off", OFF, "c32nb", B, "f32"),
            "off2": _run(tmp, "off2", OFF, "c32nb", B, "f32")}
    keys = [k for k in runs["off"].files if k.startswith("g/")]
    rms = np.sqrt(sum(float((runs["off"][k].astype(np.float64) ** 2).sum()) for k in keys) / sum(runs["off"][k].size for k in keys))
    def errs(a, b):
        return np.array([np.linalg.norm((a[k].astype(np.float64) - b[k]).ravel()) /
                         max(np.linalg.norm(b[k].astype(np.float64).ravel()), 0.1 * rms * np.sqrt(b[k].size)) for k in keys])
    e1, e0 = errs(runs["on"], runs["off"]), errs(runs["off2"], runs["off"])
    print("batch %d: fused vs separate median %.2e p90 %.2e max %.2e | separate vs separate (run to run) median %.2e p90 %.2e max %.2e"
          % (B, np.median(e1), np.percentile(e1, 90), e1.max(), np.median(e0), np.percentile(e0, 90), e0.max()))
