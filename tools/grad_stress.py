#!/usr/bin/env python3
"""Hunt an intermittent gradient glitch: one process, the same injected forward + backward repeated N times; every
repeat's gradients are compared with the first repeat's.  Float-atomic noise is ~1e-6 relative; anything above 1e-4
on a tensor that is not structurally tiny is reported with the tensors it hit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c64nb"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    from tests.common import COMPILE, engine_args, make_inputs
    from multiscale_variational_autoencoder_amd.engine import Engine
    io = make_inputs(name, B)
    eng = Engine(**engine_args(name, B)).bind()
    eng.set_params(io["params"]); eng.set_state(io["state"])
    d = {k: eng.to_device(io[k]) for k in ("x", "eps", "noise", "keep")}
    refwd = os.environ.get("STRESS_REFWD", "1") == "1"
    base, hits = None, []
    P = eng.P
    watch = [k[:-len(".conv0.w")] for k in eng.param_table if k.endswith(".mn.conv0.w")]
    def snap():
        out = {}
        for blk in watch:
            for t in ("gap", "s0", "ulin", "g", "xhat"):
                out[blk + "." + t] = eng.tensor(blk + "." + t, B).clone()
        for nm in ("dec0.b3.mn.t0", "dec0.b3.mn.t1", "dec0.b3.mn.out", "dec0.b2.mn.out", "dec0.b4.mn.t0", "dec0.b2.mn.t0"):
            out[nm] = eng.tensor(nm, B).clone()
        return out
    base_s = None
    for it in range(N):
        if it == 0 or refwd:
            eng.forward(d["x"], True, d["eps"], d["noise"], d["keep"], outputs=())
        eng.backward(COMPILE["r_loss_factor"], COMPILE["kl_loss_factor"])
        eng.sync()
        g = eng.reduce[:P].clone()
        cur_s = snap()
        if base is None:
            base = g
            base_s = cur_s
            continue
        diff = (g - base)
        bad = {}
        for k, meta in eng.param_table.items():
            n = int(np.prod(meta["shape"]))
            o = meta["offset"]
            nb = float(base[o:o + n].norm())
            if nb < 1e-3:
                continue
            e = float(diff[o:o + n].norm()) / nb
            if e > 1e-4:
                bad[k] = e
        bad = {k: v for k, v in bad.items() if (".conv" in k or ".dw." in k) and k.endswith(".w")}
        if it < 6 or bad:
            for nm in ("dec0.b3.mn.t0", "dec0.b3.mn.t1", "dec0.b3.mn.out", "dec0.b2.mn.out", "dec0.b4.mn.t0", "dec0.b2.mn.t0"):
                a, b_ = base_s[nm], cur_s[nm]
                fl = ((a > 0) != (b_ > 0))
                print("   rep %d %s %-16s maxdiff %.3e  sign flips %d  (|vals| at flips max %.3e)" % (
                    it, "GLITCH" if bad else "ok    ", nm, float((a - b_).abs().max()), int(fl.sum()),
                    float(max(a[fl].abs().max(), b_[fl].abs().max())) if fl.any() else 0.0), flush=True)
        if bad:
            hits.append((it, bad))
            print("repeat", it, "glitch:", json.dumps(dict(sorted(bad.items(), key=lambda kv: -kv[1])[:6])), flush=True)
            for k in cur_s:
                a, b_ = base_s[k], cur_s[k]
                dmax = float((a - b_).abs().max())
                note = ""
                if k.endswith(".s0"):
                    fl = ((a > 0) != (b_ > 0)).nonzero()
                    if len(fl):
                        note = " RELU FLIPS at %s values %s / %s" % (fl.tolist(), a[(a > 0) != (b_ > 0)].tolist(), b_[(a > 0) != (b_ > 0)].tolist())
                if k.endswith(".ulin"):
                    m = ((a.abs() <= 2.5) != (b_.abs() <= 2.5))
                    if m.any():
                        note = " HSIG FLIPS at %s values %s / %s" % (m.nonzero().tolist(), a[m].tolist(), b_[m].tolist())
                if note or dmax > 1e-5:
                    print("     ", k, "max abs diff %.3e" % dmax, note, flush=True)
    print("variant", {k: v for k, v in os.environ.items() if k.startswith("MVAE_") or k.startswith("STRESS_")},
          "repeats", N, "glitches", len(hits), flush=True)


if __name__ == "__main__":
    main()
