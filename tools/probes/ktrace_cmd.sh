#!/bin/bash
# kernel trace of an arbitrary python command line (relative to the repo root): per kernel name launches, mean and total time
#   usage: ktrace_cmd.sh <N divide-by (e.g. traced iterations)> python-script args...
R="$(cd "$(dirname "$0")/../.." && pwd)"
DIV=$1; shift
cd /tmp && export TMPDIR=/tmp
d=/tmp/ktc_$$; rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 "$R/$1" "${@:2}" > /tmp/ktc_$$.log 2>&1
python3 - "$d" "$DIV" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
div = float(sys.argv[2])
acc = collections.defaultdict(list)
rows = list(csv.DictReader(open(f)))
for r in rows:
    acc[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print("all kernels: %d launches, %.3f ms  (/%g: %.1f launches, %.3f ms)" % (len(rows), tot / 1e3, div, len(rows) / div, tot / 1e3 / div))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print("%8.1f /it %7.2f us avg %9.3f ms/it  %s" % (len(v) / div, sum(v) / len(v), sum(v) / 1e3 / div, k[:100]))
PY
