"""Compile libmvae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libmvae_hip.so")
SOURCES = ["kernels_generic.hip", "kernels_mfma.hip", "kernels_opt.hip", "kernels_edge.hip", "kernels_dw.hip", "kernels_dense.hip", "kernels_se.hip", "kernels_bf16.hip", "kernels_split.hip", "kernels_split_wgrad.hip", "kernels_fused.hip", "kernels_fused_fwd.hip", "kernels_blocks.hip", "kernels_layers.hip", "dispatch.hip", "layer_ops.cpp", "runtime.cpp"]
HEADERS = ["kernels.h", "act16.h", "prof.h", "split.h", os.path.join("..", "..", "include", "mvae_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmvae_hip.so cannot be built")


STAMP_PATH = os.path.join(CSRC, ".build_flags")


def _flag_set():
    """The environment switches that change the generated code: a library built under one set must not survive as another."""
    return "packed_f32=%s debug=%s" % (os.environ.get("MVAE_PACKED_F32", "") == "1", os.environ.get("MVAE_DEBUG_BUILD", "") == "1")


def _flags_changed():
    try:
        return open(STAMP_PATH).read().strip() != _flag_set()
    except OSError:
        # no stamp: a prebuilt library that travelled without it (the GPU box) counts as built with the default flags
        return os.path.exists(LIB_PATH) and _flag_set() != "packed_f32=False debug=False"


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC over csrc/*.hip + runtime.cpp -> libmvae_hip.so."""
    if not force and _flags_changed():
        force = True
    if not force and not _stale():
        return LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, jobs = [], []
    for src in srcs:
        obj = os.path.splitext(src)[0] + ".o"
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                os.path.getmtime(src), *[os.path.getmtime(os.path.join(CSRC, hd)) for hd in HEADERS]):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", obj]
            # No packed-float32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel: on MI355X a
            # v_pk_fma_f32 of one wave returns wrong values now and then while another wave of the same SIMD interleaves
            # float32 VALU work with v_mfma_f32_32x32x16_bf16 -- which is what the split-bf16 convolutions do
            # (csrc/kernels_split.hip, "Hardware self-test").  MVAE_PACKED_F32=1 builds with them.
            if os.environ.get("MVAE_PACKED_F32", "") != "1":
                cmd[1:1] = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Wno-unknown-attributes"]
            if os.environ.get("MVAE_DEBUG_BUILD", "") == "1":     # timing-diagnostic switches (runtime.cpp); never the default
                cmd.insert(1, "-DMVAE_DEBUG_BUILD")
            jobs.append(cmd)
        objs.append(obj)
    if jobs:
        # one hipcc per translation unit, a few at a time (a unit peaks at ~2 GB; MVAE_BUILD_JOBS overrides)
        from concurrent.futures import ThreadPoolExecutor
        nj = max(1, int(os.environ.get("MVAE_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1))

        def _one(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        with ThreadPoolExecutor(nj) as ex:
            list(ex.map(_one, jobs))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(STAMP_PATH, "w") as f:
        f.write(_flag_set() + "\n")
    return LIB_PATH


def build_tools(verbose=False):
    """tools/bf16_unit.bin: every bf16 MFMA kernel alone against a double-precision reference (tests/test_bf16_kernels_gpu.py
    runs it on the GPU box; built here because the box has no reason to have a compiler warmed up)."""
    root = os.path.dirname(HERE)
    if os.environ.get("MVAE_BUILD_PROBES", "") == "1":     # timing probes linked against the same objects (not run by tests)
        for name in ("dual_probe.cpp", "conv_probe_s.cpp"):
            _build_tool(os.path.join(root, "tools", name), verbose)
    return _build_tool(os.path.join(root, "tools", "bf16_unit.hip"), verbose)


def _build_tool(src, verbose=False):
    out = os.path.splitext(src)[0] + ".bin"
    objs = [os.path.join(CSRC, os.path.splitext(s)[0] + ".o") for s in SOURCES if s != "runtime.cpp"]
    deps = [src] + objs
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps if os.path.exists(d)):
        return out
    quiet = None if verbose else subprocess.DEVNULL
    tobj = os.path.splitext(out)[0] + ".o"
    for cmd in ([_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", "-c", src, "-o", tobj],
                [_hipcc(), "--offload-arch=gfx950", tobj] + objs + ["-o", out]):
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, stdout=quiet, stderr=quiet)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
