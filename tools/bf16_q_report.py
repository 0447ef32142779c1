"""Print the bf16 parity reports (gpurun_out/parity_bf16_*_b*.json): float64 oracle vs rounding-aware oracle."""
import json, glob
for f in sorted(glob.glob('gpurun_out/parity_bf16_*_b*.json')):
    r = json.load(open(f)); q = r["q"]
    print(f.split('/')[-1])
    for name, d in (("f64", r), ("q", q)):
        print("  %-3s elbo %.2e fwd_worst %.2e (%s) fwd_med %.2e grad_worst %s vec_worst %s med %.2e p90 %.2e kink %d/%d maxd %.2e loss_flips %d" % (
            name, d["elbo_rel"], d["fwd_worst"][1], d["fwd_worst"][0], d["fwd_median"], [(k, round(v, 4)) for k, v in d["grad_worst"][:2]],
            [(k, round(v, 4)) for k, v in d["grad_vec_worst"][:2]], d["grad_median"], d["grad_p90"], d["kink"]["flips"], d["kink"]["units"],
            d["kink"]["max_abs_at_flip"], d["kink"]["loss_flips"]))
