#!/usr/bin/env python3
"""ISA audit of the HIP sources: for every kernel, how many global loads it has and how many of them are followed
within a few instructions by `s_waitcnt vmcnt(0)` before the next load ("serialised": the load's latency is exposed
instead of overlapping the next loads).  This check found the exec-masked / run-time-trip-count load chains hipcc
serialises (load scheduling rule).

    python tools/isa_audit.py            # compiles csrc/*.hip to /tmp/mvae_isa/*.s (device only) and audits them
    python tools/isa_audit.py --flow multiscale_variational_autoencoder_amd/csrc/kernels_mfma.hip k_gemm_dual
                                         # compact load / wait / MFMA / store / barrier flow of one kernel
"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "multiscale_variational_autoencoder_amd", "csrc")
OUT = "/tmp/mvae_isa"


def compile_all(files):
    os.makedirs(OUT, exist_ok=True)
    outs = []
    for f in files:
        o = os.path.join(OUT, os.path.basename(f).replace(".hip", ".s"))
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", f, "-o", o],
                       check=True, stderr=subprocess.DEVNULL)
        outs.append(o)
    return outs


def functions(path):
    txt = open(path).read()
    for fn in re.split(r"\n(?=_Z[\w]+:\s)", txt):
        m = re.match(r"(_Z\w+):", fn)
        if m:
            end = fn.find(".amdhsa_kernel")
            yield m.group(1), fn[:end if end > 0 else len(fn)]


def demangle(sym):
    return subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()


def audit(paths):
    for p in paths:
        for sym, body in functions(p):
            lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
            loads = sum(1 for l in lines if l.startswith(("global_load", "buffer_load")))
            ser = 0
            for i, l in enumerate(lines):
                if l.startswith("global_load"):
                    for k in range(1, 5):
                        if i + k < len(lines):
                            if lines[i + k].startswith("global_load"):
                                break
                            if lines[i + k].startswith("s_waitcnt vmcnt(0)"):
                                ser += 1
                                break
            if loads:
                print("%-70s loads %4d  serialised %4d" % (demangle(sym)[:70], loads, ser))


def flow(path, name):
    (s,) = compile_all([path])
    for sym, body in functions(s):
        d = demangle(sym)
        if name not in d:
            continue
        out = []
        for l in body.split("\n"):
            l = l.strip()
            if "Loop Header" in l:
                out.append("LOOP")
            elif l.startswith("global_load_dwordx4"):
                out.append("LD4")
            elif l.startswith("global_load"):
                out.append("LD1")
            elif l.startswith("global_store"):
                out.append("ST")
            elif l.startswith("global_atomic"):
                out.append("AT")
            elif l.startswith("v_mfma"):
                out.append("M")
            elif l.startswith("s_barrier"):
                out.append("BAR")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                out.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
        comp, prev, n = [], None, 0
        for t in out + [None]:
            if t == prev:
                n += 1
            else:
                if prev is not None:
                    comp.append(prev if n == 1 else "%sx%d" % (prev, n))
                prev, n = t, 1
        print(d[:100])
        print("  " + " ".join(comp))


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--flow":
        flow(sys.argv[2], sys.argv[3])
    else:
        audit(compile_all(sorted(glob.glob(os.path.join(SRC, "*.hip")))))
