"""Where the cycles of ONE conv-like layer go, from the SQ counters (run on the GPU box):

    SV_BENCH_K=3 SV_BENCH_S=2 python tools/pmc_sq.py 1024 160 32 320 fwd [kernel-name substring ...]

A few `rocprofv3 --pmc` passes (kernel trace only) over tools/layer_bench.py; prints, per matched kernel, the counters of
one launch and the ratios that matter: share of wave cycles spent waiting, VALU / LDS / VMEM / MFMA busy, LDS bank
conflict share."""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAIT_INST_ANY"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"],
          ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"],
          ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU"],
          ["SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_MFMA"]]
ITERS, WARM = 3, 1


def main():
    B, Cin, H, N = sys.argv[1:5]
    kind = sys.argv[5]
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pmc_traffic as P
    names = sys.argv[6:] or [n for n in P.ALL_KERNELS if n != "slab_reduce_kernel"]
    out = {}
    for i, ctrs in enumerate(PASSES):
        d = os.path.join(ROOT, "gpurun_out", "pmc_sq", "p%d_%d" % (os.getpid(), i))
        env = dict(os.environ, SV_BENCH_ITERS=str(ITERS), SV_BENCH_WARM=str(WARM))
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "layer_bench.py"), B, Cin, H, N, kind]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
        if r.returncode != 0:
            print("pass %d failed: %s" % (i, r.stderr[-600:]))
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for row in csv.DictReader(open(files[0])):
            nm = P.kernel_of(row["Kernel_Name"], names)
            if nm:
                out.setdefault(nm, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for nm, cs in out.items():
        v = {k: sum(x[-ITERS:]) / ITERS for k, x in cs.items()}       # mean of the timed launches (last ITERS dispatches)
        print("== %s  (B=%s Cin=%s H=%s N=%s %s k=%s s=%s)" % (nm, B, Cin, H, N, kind, os.environ.get("SV_BENCH_K", "3"),
                                                             os.environ.get("SV_BENCH_S", "1")))
        for k in sorted(v):
            print("  %-28s %16.0f" % (k, v[k]))
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc:
            for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"):
                if k in v:
                    print("  %-28s / WAVE_CYCLES = %.3f" % (k, v[k] / wc))
        if v.get("SQ_BUSY_CYCLES"):
            print("  MFMA_BUSY / BUSY_CYCLES = %.3f" % (v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / v["SQ_BUSY_CYCLES"]))
        if v.get("SQ_LDS_IDX_ACTIVE"):
            print("  LDS_BANK_CONFLICT / LDS_IDX_ACTIVE = %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))


if __name__ == "__main__":
    main()
