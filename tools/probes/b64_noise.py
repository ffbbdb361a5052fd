"""Distribution of the bf16 loss-scalar errors of the B = 64 step against the fp32 oracle on the atomic path (tests/test_model_gpu.py
_b64_run): N runs, per scalar the sorted errors.  SV_LIB_PATH selects the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_model_gpu as M
from tests import _cases as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
runs = [M._b64_run("bf16") for _ in range(N)]
for k in T.SCALARS:
    e = sorted(r["scalar"][k] for r in runs)
    print("%-14s median %.2e  max %.2e   %s" % (k, e[len(e) // 2], e[-1], " ".join("%.1e" % v for v in e)))
print("cos", sorted("%.4f" % r["cos"] for r in runs))
