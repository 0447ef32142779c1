#!/usr/bin/env python3
"""Compact instruction flow of one kernel (as the library is compiled): global loads / stores / atomics, LDS reads / writes,
MFMAs, barriers, waits, and the number of VALU / SALU instructions between them.
    python tools/isa_flow.py kernels_split.hip k_gemm_dual_sILi1        # file under csrc/, substring of the MANGLED name
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "multiscale_variational_autoencoder_amd", "csrc", sys.argv[1])
out = "/tmp/mvae_isa/%s.s" % os.path.basename(src)
os.makedirs("/tmp/mvae_isa", exist_ok=True)
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
                "-S", "--cuda-device-only", src, "-o", out] + sys.argv[3:], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for fn in re.split(r"\n(?=_Z[\w]+:\s)", txt):
    m = re.match(r"(_Z\w+):", fn)
    if not m or sys.argv[2] not in m.group(1):
        continue
    body = fn[:fn.find(".amdhsa_kernel")] if ".amdhsa_kernel" in fn else fn
    toks, v, sa = [], 0, 0
    def flush():
        global v, sa
        if v or sa:
            toks.append("v%d" % v + (",s%d" % sa if sa else ""))
        v = sa = 0
    for l in body.split("\n"):
        l = l.strip()
        if not l or l.startswith(";"):
            continue
        op = l.split()[0]
        t = None
        if re.match(r"\.LBB\d+_\d+:", l): t = "\n" + l
        elif op.startswith("global_load") or op.startswith("buffer_load"): t = "LD" + ("4" if "x4" in op else "1")
        elif op.startswith("global_store") or op.startswith("buffer_store"): t = "ST"
        elif op.startswith("global_atomic"): t = "AT"
        elif op.startswith("ds_read_b64_tr") : t = "Rt"
        elif op.startswith("ds_read") or op.startswith("ds_load"): t = "R" + op.split("_b")[-1]
        elif op.startswith("ds_write") or op.startswith("ds_store"): t = "Wr" + op.split("_b")[-1]
        elif op.startswith("v_mfma"): t = "M"
        elif op.startswith("s_barrier"): t = "BAR"
        elif op.startswith("s_waitcnt"):
            t = "wait(" + ",".join(re.findall(r"(vmcnt\(\d+\)|lgkmcnt\(\d+\))", l)).replace("cnt", "") + ")"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): t = op + "->" + l.split()[-1]
        elif op.startswith("v_"): v += 1; continue
        elif op.startswith("s_"): sa += 1; continue
        else: continue
        flush(); toks.append(t)
    flush()
    comp, prev, n = [], None, 0
    for t in toks + [None]:
        if t == prev: n += 1
        else:
            if prev is not None: comp.append(prev if n == 1 else "%sx%d" % (prev, n))
            prev, n = t, 1
    print(m.group(1)); print(" ".join(comp))
