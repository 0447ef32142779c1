#!/usr/bin/env python3
"""Device-side time of the stand-alone Laplacian pyramid (mvae_laplacian_split / _merge), inputs resident in HBM:
python tools/lap_bench.py  ->  per call: us, GB/s of algorithmic bytes (split: read x, write the levels; merge: read the
levels, write the image; intermediates not counted)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiscale_variational_autoencoder_amd import _abi, layer_blocks as lb  # noqa: E402

lib = _abi.load_library()
dev = torch.device("cuda", 0)
for (b, h, w, c, levels) in [(512, 32, 32, 3, 3), (64, 256, 256, 3, 7), (512, 256, 256, 3, 3)]:
    x = torch.rand((b, h, w, c), device=dev) * 255.0
    outs = [torch.empty((b, h >> i, w >> i, c), device=dev) for i in range(levels)]
    work = torch.empty(2 * b * h * w * c + 16, device=dev)
    back = torch.empty_like(x)
    g = (C.c_float * 9)(*np.asarray(lb.gaussian_kernel((3, 3), (1, 1)), np.float32).ravel())
    ptrs = (C.c_void_p * levels)(*[o.data_ptr() for o in outs])
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def split():
        assert lib.mvae_laplacian_split(0, C.c_void_p(x.data_ptr()), b, h, w, c, levels, 0.0, 255.0, g, ptrs,
                                        C.c_void_p(work.data_ptr()), st) == 0

    def merge():
        assert lib.mvae_laplacian_merge(0, ptrs, b, h, w, c, levels, 0.0, 255.0, C.c_void_p(back.data_ptr()),
                                        C.c_void_p(work.data_ptr()), st) == 0

    nbytes = 4.0 * b * h * w * c * (1.0 + sum(0.25 ** i for i in range(levels)))
    for name, fn in (("split", split), ("merge", merge)):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / 20
        print("%-5s B=%d %dx%dx%d levels=%d : %8.1f us  %7.1f GB/s" % (name, b, h, w, c, levels, us, nbytes / us / 1e3))
    print("      round trip max |err| = %.3g" % float((back - x).abs().max()))
