"""A/B of MVAE_LSB_MASK on the GRADIENT error of the bf16 path: 1 = dt2 rounded to seven significant
bits with the ReLU mask of t1 in its LSB (the depthwise backward does not read t1: 3 tensor passes instead of 4), 0 = eight
bits and t1 is read.  Error against the float64 oracle (what the storage format costs) and against the rounding-aware oracle
(what the kernels add).  Writes profiles/round3_lsb_mask_ab.json."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_bf16_gpu as T

out = {}
for name, B, nbf in (("c64nb", 32, 3), ("c256nb", 2, 5)):
    for lsb in ("1", "0"):
        os.environ["MVAE_LSB_MASK"] = lsb
        rep, eng = T._report(name, B, nbf)
        eng.close()
        row = {}
        for tag, d in (("vs_float64", rep), ("vs_rounding_aware", rep["q"])):
            row[tag] = dict(elbo_rel=d["elbo_rel"], grad_median=d["grad_median"], grad_p90=d["grad_p90"],
                            worst_weight=d["grad_worst"][0], worst_vector=d["grad_vec_worst"][0], fwd_worst=d["fwd_worst"])
        out["%s_b%d_lsb%s" % (name, B, lsb)] = row
        print(name, B, "LSB_MASK=" + lsb, json.dumps(row["vs_float64"]), flush=True)
os.environ.pop("MVAE_LSB_MASK", None)
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(os.path.join("gpurun_out", "round3_lsb_mask_ab.json"), "w"), indent=1)
