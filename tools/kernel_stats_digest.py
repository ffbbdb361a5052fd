"""Readable digest of a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv.

    python tools/kernel_stats_digest.py <kernel_stats.csv> <executed steps> "<header line>" > digest.txt"""
import csv
import sys

path, steps, header = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print("# " + header)
print("# sum of kernel durations %.2f ms/step over %d executed steps" % (total / steps / 1e6, steps))
print("%-78s %10s %10s %10s %6s" % ("kernel", "calls/step", "avg us", "ms/step", "%"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    t = float(r["TotalDurationNs"])
    print("%-78s %10.1f %10.1f %10.3f %6.1f" % (r["Name"][:78], float(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                               t / steps / 1e6, 100.0 * t / total))
