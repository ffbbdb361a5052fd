"""Where the wall time of one step goes: kernel trace of bench.py (default schedule), busy / idle time of the GPU per step and
the largest idle gaps with the kernels around them.  Run on the GPU box from the repo root:

    python tools/step_timeline.py [extra bench.py args]

(rocprofv3 --kernel-trace; this process never touches the GPU)"""
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = "/tmp/sv_timeline"
subprocess.run(["rm", "-rf", d])
steps, warm = 6, 4
cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
       "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline", "--no-roofline", "--no-extras"] + sys.argv[1:]
r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
print(r.stdout[-400:])
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, r.stderr[-2000:]
rows = []
for row in csv.DictReader(open(files[0])):
    rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"], row.get("Stream_Id", row.get("Queue_Id", "?"))))
rows.sort()
# one step = from one sgd_kernel to the next
sgd = [i for i, r_ in enumerate(rows) if "sgd_kernel" in r_[2]]
assert len(sgd) >= 3
a, b = sgd[-3], sgd[-2]                      # a full step between two SGD launches (steady state)
seg = rows[a + 1: b + 1]
t0, t1 = rows[a][1], rows[b][1]
print("step wall (sgd end -> sgd end): %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
gaps = []
prev_end, prev_name = t0, rows[a][2]
for s, e, n, q in seg:
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    if s > prev_end:
        gaps.append((s - prev_end, prev_name[:60], n[:60]))
    if e > prev_end:
        prev_end, prev_name = e, n
busy += cur_e - cur_s
print("GPU busy (union of kernel intervals) %.3f ms, idle %.3f ms, sum of kernel durations %.3f ms" % (
    busy / 1e6, (t1 - t0 - busy) / 1e6, sum(e - s for s, e, _, _ in seg) / 1e6))
gaps.sort(reverse=True)
print("idle gaps: %d, total %.3f ms; > 5 us: %d (%.3f ms)" % (len(gaps), sum(g[0] for g in gaps) / 1e6,
      sum(1 for g in gaps if g[0] > 5000), sum(g[0] for g in gaps if g[0] > 5000) / 1e6))
for gp, pn, nn in gaps[:25]:
    print("  %7.1f us   after %-60s before %s" % (gp / 1e3, pn, nn))
# time by kernel family inside the step
fam = {}
for s, e, n, q in seg:
    k = n.split("(")[0].split("<")[0][-40:]
    fam[k] = fam.get(k, 0) + (e - s)
for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-42s %8.3f ms" % (k, v / 1e6))
