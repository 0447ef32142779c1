#!/bin/bash
# rocprofv3 kernel trace of a few headline steps + per-queue analysis of the last one (tools/trace_queues.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; O=gpurun_out/r3/trace1; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-profile > $O/bench.json 2> $O/bench.err
echo "rc=$?"; python tools/trace_queues.py $O > gpurun_out/r3/trace1_queues.txt; head -50 gpurun_out/r3/trace1_queues.txt
find $O -name "*.csv" -size +30M -delete
