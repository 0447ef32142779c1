"""Per-tensor forward errors of the bf16 path against the rounding-aware oracle, in network order (debug aid)."""
import json, sys
r = json.load(open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/parity_bf16_c32nb_b8.json'))
fa = r["q"]["fwd_all"]; fe = r.get("fwd_all", {})
order = []
for s in range(3):
    order += ["enc%d.conv_base" % s] + [x for i in range(5) for x in (["enc%d.b%d.conv" % (s, i)] if i in (0, 4) else []) + ["enc%d.b%d.mn.%s" % (s, i, t) for t in ("t0", "t1", "out")]]
    order += ["dec%d.dense" % s] + [x for i in range(5) for x in (["dec%d.b%d.convT" % (s, i)] if i in (0, 4) else []) + ["dec%d.b%d.mn.%s" % (s, i, t) for t in ("t0", "t1", "out")]]
for k in order:
    if k in fa and k.startswith(("enc0", "dec0")):
        print("%-22s q %.2e   f64 %.2e" % (k, fa[k], fe.get(k, float('nan'))))
print(r["q"]["grad_worst"][:4], r["q"]["grad_vec_worst"][:3])
