#!/bin/bash
# the two bench lines that go into profiles/ (default run, then BASELINE config 4); GPU box, repo root
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
timeout -k 10 500 python bench.py > gpurun_out/r3/bench_final_c32nb.json 2> gpurun_out/r3/bench_final_c32nb.err || exit 1
timeout -k 10 300 python bench.py --workload c256nb > gpurun_out/r3/bench_final_c256nb.json 2> gpurun_out/r3/bench_final_c256nb.err || exit 1
python - <<'PY'
import json
for wl in ("c32nb", "c256nb"):
    d = json.loads(open("gpurun_out/r3/bench_final_%s.json" % wl).read().strip().splitlines()[-1])
    print(wl, "%.3f ms" % d["ms_per_step"], "%.0f img/s" % d["value"], "hbm_frac %.4f" % d["step_roofline"]["hbm_frac"], d["roofline"]["kernel"], "%.3f" % d["roofline"]["frac"], d["roofline"]["traffic"])
PY
