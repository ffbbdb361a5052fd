#!/bin/bash
# kernel sequence of ONE eager config-5 iteration, in launch order with durations (rocprofv3 --kernel-trace)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
d=/tmp/kt_seq; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 "$R/tools/probes/svhn_layers.py" ${1:-1024} 4 > /tmp/kt_seq.log 2>&1
python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last iteration: from the last-but-one adam_kernel to the last
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
seq = rows[idx[-2] + 1: idx[-1] + 1]
tot = 0.0
for r in seq:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += us
    print("%8.2f us  grid %-8s %s" % (us, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:110]))
print("%d kernels, %.3f ms" % (len(seq), tot / 1e3))
PY
