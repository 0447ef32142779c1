#!/bin/bash
# Kernel trace of a few headline steps -> per-stream (queue) kernel time table: which kernels make up the longest chain.
#   OUT=gpurun_out/r4/trace tools/trace_stream0.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=${OUT:-gpurun_out/r4/trace}; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-secondary --no-kernel-profile "$@" > $O/bench.json 2> $O/bench.err
echo "trace rc=$?"
python3 tools/trace_streams.py $O/t > $O/summary.txt 2>&1
head -80 $O/summary.txt
find $O -name "*.csv" -size +30M -delete
