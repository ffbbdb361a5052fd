#!/bin/bash
# Regenerates the artifacts of profiles/ on the GPU box into gpurun_out/profiles_new/ (copy what should be judged into
# profiles/ afterwards).  usage: bash tools/refresh_profiles.sh rNN [pmc]
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-r02}
OUT="$R/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 30 --warmup 10 2>/dev/null | tail -1 > "$OUT/${TAG}_bench.json"
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/prof_b.log 2>&1
F=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_bench_kernel_stats.csv"
python3 "$R/tools/kernel_stats_digest.py" "$F" 15 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline   (WRN-28-2, K=10, B_l=B_u=512, bf16; grouped schedule, eager, weight gradients on a side stream; $(grep "^{.metric" /tmp/prof_b.log | tail -1 | python3 -c 'import sys,json; print("wall %.2f ms/step under the profiler" % json.loads(sys.stdin.read())["ms_per_step"])'))" > "$OUT/${TAG}_bench_kernel_stats.txt"
{ python3 "$R/tools/layer_bench.py"; for s in "2048 32 32 32" "2048 64 16 64" "2048 128 8 128"; do python3 "$R/tools/layer_bench.py" $s; done; } 2>/dev/null | grep "of bf16" > "$OUT/${TAG}_layer_bench_wrn28_10.txt"
python3 "$R/bench.py" --net wideresnet-28-10 --batch 256 --classes 100 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_cfg4_wrn28_10_bench.json"
rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- python3 "$R/bench.py" --net wideresnet-28-10 --batch 256 --classes 100 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_c.log 2>&1
F=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_cfg4_wrn28_10_kernel_stats.csv"
python3 "$R/tools/kernel_stats_digest.py" "$F" 10 "rocprofv3 --kernel-trace --stats -- python3 bench.py --net wideresnet-28-10 --batch 256 --classes 100 --steps 6 --warmup 2   (BASELINE config 4; grouped schedule, eager)" > "$OUT/${TAG}_cfg4_wrn28_10_kernel_stats.txt"
if [ "$2" = "pmc" ]; then python3 "$R/tools/pmc_traffic.py" "$OUT/${TAG}_pmc_traffic.json" > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log; fi
ls -la "$OUT"
