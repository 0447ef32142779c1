"""Run-to-run spread of the default mode next to the default-vs-deterministic difference, per gradient tensor (c64nb, batch 4:
the configuration of tests/test_deterministic_gpu.py).  A tensor whose two DEFAULT runs differ by as much as default differs
from deterministic is summation-order noise (terms that cancel), not a kernel difference."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.test_deterministic_gpu import _steps
from tests.common import rel_err
name, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("c64nb", 4)
d = _steps(name, B, True)
c1 = _steps(name, B, False); c2 = _steps(name, B, False); c3 = _steps(name, B, False)
gmax = max(np.linalg.norm(v) for v in d["grads"].values())
rows = []
for k in d["grads"]:
    if np.linalg.norm(d["grads"][k]) > 1e-3 * gmax:
        rows.append((max(rel_err(c["grads"][k], d["grads"][k]) for c in (c1, c2, c3)),
                     max(rel_err(c1["grads"][k], c2["grads"][k]), rel_err(c1["grads"][k], c3["grads"][k]), rel_err(c2["grads"][k], c3["grads"][k])), k,
                     float(np.linalg.norm(d["grads"][k]) / gmax)))
rows.sort(reverse=True)
for r in rows[:12]:
    print("default vs det %.2e   default vs default %.2e   |g|/gmax %.1e   %s" % (r[0], r[1], r[3], r[2]))
