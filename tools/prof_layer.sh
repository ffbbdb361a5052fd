#!/bin/bash
# rocprofv3 kernel trace of one layer micro-benchmark: per-kernel average durations.  usage: prof_layer.sh B C H N
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_layer
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_layer -o lb -- python3 "$R/tools/layer_bench.py" "$@" > /tmp/prof_layer.log 2>&1
python3 - <<'PY'
import csv, glob
fs = glob.glob("/tmp/prof_layer/**/*kernel_stats.csv", recursive=True)
if not fs:
    print("no stats file:", glob.glob("/tmp/prof_layer/**/*", recursive=True)[:10]); print(open("/tmp/prof_layer.log").read()[-2000:])
else:
    for r in list(csv.DictReader(open(fs[0])))[:10]:
        print("%-100s calls %6s avg %9.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
