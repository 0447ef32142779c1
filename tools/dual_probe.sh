#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
for v in ${SD:-0 1}; do for hw in 32 16 8; do
  echo -n "SPLIT_DUAL=$v H=$hw: "; MVAE_SPLIT_DUAL=$v MVAE_SPLIT_SELFTEST=0 timeout -k 10 60 ./tools/dual_probe.bin 512 $hw || exit 1
done; done 2>&1 | cut -c1-120 | tee gpurun_out/r3/dual_probe.log
[ -n "$VARIANTS" ] && VARIANTS="$VARIANTS" bash tools/dual_variants.sh | cut -c1-120
exit 0
