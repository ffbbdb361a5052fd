"""PMC traffic of ONE conv-like layer (run on the GPU box):  SV_BENCH_K=1 SV_BENCH_S=2 python tools/pmc_layer.py B Cin H N kind
Same two counter passes and the same formula as tools/pmc_traffic.py; the launch's kernels are matched against pmc_traffic.ALL_KERNELS
(every conv-like kernel of the library) unless kernel-name substrings follow `kind`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pmc_traffic as P      # noqa: E402

B, Cin, H, N = map(int, sys.argv[1:5])
kind = sys.argv[5]
names = sys.argv[6:] or P.ALL_KERNELS
TAG = "probe_%d_%d_%d_%d_%s_k%s_s%s" % (B, Cin, H, N, kind, os.environ.get("SV_BENCH_K", "3"), os.environ.get("SV_BENCH_S", "1"))
P.LAYERS[TAG] = (B, Cin, H, N, kind, names)
outdir = os.path.join(P.ROOT, "gpurun_out", "pmc")
os.makedirs(outdir, exist_ok=True)
f, k = P.one_pass(TAG, "FETCH_SIZE", outdir)
w, _ = P.one_pass(TAG, "WRITE_SIZE", outdir)
print("B=%d Cin=%d H=%d N=%d %s k=%s s=%s kernels=%s: FETCH %.1f MiB (x2 = %.1f MB read)  WRITE %.1f MB  traffic %.1f MB" % (
    B, Cin, H, N, kind, os.environ.get("SV_BENCH_K", "3"), os.environ.get("SV_BENCH_S", "1"), k, f / 1024,
    2 * f * 1024 / 1e6, w * 1024 / 1e6, (2 * f + w) * 1024 / 1e6), flush=True)
