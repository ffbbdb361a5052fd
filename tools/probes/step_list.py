"""Every kernel of one steady-state step of the default bench, in start order: start offset, duration, gap to the previous
kernel's end on the same queue, short name.  Run on the GPU box:  python tools/probes/step_list.py > out.txt"""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = "/tmp/sv_steplist"
subprocess.run(["rm", "-rf", d])
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
       "--steps", "6", "--warmup", "4", "--no-cpu-baseline", "--no-roofline", "--no-extras"] + sys.argv[1:]
r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, r.stderr[-2000:]
rows = []
for row in csv.DictReader(open(files[0])):
    rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"], row.get("Queue_Id", "?")))
rows.sort()
sgd = [i for i, r_ in enumerate(rows) if "sgd_kernel" in r_[2]]
a, b = sgd[-3], sgd[-2]
seg = rows[a: b + 1]
t0 = seg[0][1]
last_end = {}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)", n)
    if m:
        return m.group(1)
    n = n.split("(")[0]
    n = re.sub(r"at::native::", "", n)
    return n[:60]


print("step wall %.3f ms, %d kernels" % ((seg[-1][1] - t0) / 1e6, len(seg) - 1))
small = 0.0
for s, e, n, q in seg[1:]:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    dur = (e - s) / 1e3
    print("%9.1f us  q%-3s dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, q, dur, gap, short(n)))
