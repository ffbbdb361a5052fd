#!/bin/bash
# kernel-trace of one layer_bench invocation: average duration per kernel name (us).   usage: ktrace.sh <layer_bench args...>
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd /tmp && export TMPDIR=/tmp
d=/tmp/kt_$$; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 "$R/tools/layer_bench.py" "$@" > /dev/null 2>&1
python3 - "$d" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        v = v[3:]
        print("%9.1f us x %3d  %s" % (sum(v) / len(v), len(v), k[:110]))
PY
