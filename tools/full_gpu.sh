#!/bin/bash
# full GPU suite + the two bench lines (round-end rehearsal); run on the GPU box from the repo root
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3/full_suite.log 2>&1
rc=$?; tail -5 gpurun_out/r3/full_suite.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload c256nb > gpurun_out/r3/bench_final_c256nb.json 2> gpurun_out/r3/bench_final_c256nb.err
rc=$?; tail -c 600 gpurun_out/r3/bench_final_c256nb.json
exit $rc
