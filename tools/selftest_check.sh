#!/bin/bash
# does the bind-time self-test (kernels_split.hip) agree with the ground truth (tools/split_debug2.py) on this box?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3
rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1
for i in 1 2 3; do
MVAE_SPLIT_SELFTEST=2 timeout -k 10 100 python - <<'PY' 2>&1 | grep -E "self-test|status|took"
import time, sys, os
sys.path.insert(0, os.getcwd())
from tests.common import engine_args
from multiscale_variational_autoencoder_amd.engine import Engine
import torch
torch.cuda.init()
t0 = time.time()
eng = Engine(**engine_args("c32nb", 8)).bind(0)
print("status", eng.lib.mvae_split_conv_status(), "bind took %.3f s" % (time.time() - t0))
PY
done
echo "TRUTH: $(MVAE_SPLIT_SELFTEST=0 timeout -k 10 200 python tools/split_debug2.py c256nb 2 40 2>&1 | grep 'split 1 reps')"
