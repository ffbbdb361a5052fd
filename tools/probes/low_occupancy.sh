# kernels of one step whose grid gives fewer than two workgroups per CU (512) and that run longer than 20 us
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lo
rocprofv3 --kernel-trace --output-format csv -d /tmp/lo -o t -- python3 /root/repo/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --wgrad-side 0 $SV_LO_ARGS > /dev/null 2>&1
F=$(find /tmp/lo -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0])
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1)
    nwg = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(wg, 1)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (r["Kernel_Name"][:70], nwg, wg)
    agg[k][0] += 1; agg[k][1] += d
lim = int(__import__("os").environ.get("SV_LO_WGS", "512"))
out = [(v[1] / v[0], v[0], k) for k, v in agg.items() if k[1] < lim and v[1] / v[0] > 20]
for avg, n, k in sorted(out, reverse=True)[:40]:
    print("%7.1f us x%4d  wgs %5d x %3d  %s" % (avg, n, k[1], k[2], k[0]))
PY
