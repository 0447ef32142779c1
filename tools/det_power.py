import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_deterministic_gpu import _steps
for det in (False, True):
    a = _steps("c32nb", 16, det); b = _steps("c32nb", 16, det)
    nd = sum(0 if np.array_equal(a["grads"][k], b["grads"][k]) else 1 for k in a["grads"])
    print("det", det, "gradient tensors that differ between two runs:", nd, "of", len(a["grads"]), "recon equal", np.array_equal(a["recon"], b["recon"]))
