#!/bin/bash
# Regenerates the artifacts of profiles/ on the GPU box into gpurun_out/profiles_new/ (copy what should be judged into
# profiles/ afterwards).  usage: bash tools/refresh_profiles.sh rNN [pmc]
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=${1:-r03}
PMC=$2
OUT="$R/gpurun_out/profiles_new"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 30 --warmup 10 2>/dev/null | tail -1 > "$OUT/${TAG}_bench.json"
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o b -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --wgrad-side 0 > /tmp/prof_b.log 2>&1
F=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_bench_kernel_stats.csv"
python3 "$R/tools/kernel_stats_digest.py" "$F" 13 "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extras --wgrad-side 0   (WRN-28-2, K=10, B_l=B_u=512, bf16; grouped schedule, eager, ONE stream so that per-kernel durations are not inflated by the side stream's concurrent weight gradients; $(grep "^{.metric" /tmp/prof_b.log | tail -1 | python3 -c 'import sys,json; print("wall %.2f ms/step under the profiler" % json.loads(sys.stdin.read())["ms_per_step"])'))" > "$OUT/${TAG}_bench_kernel_stats.txt"
{ python3 "$R/tools/layer_bench.py"; for s in "2048 32 32 32" "2048 64 16 64" "2048 128 8 128"; do python3 "$R/tools/layer_bench.py" $s; done; } 2>/dev/null | grep "of bf16" > "$OUT/${TAG}_layer_bench_wrn28_10.txt"
# the WRN-28-10 odd layers (stride-2 3x3, 1x1 shortcuts, first block) at 4 x 256 images: SV_BENCH_K / SV_BENCH_S
{ for s in "3 2 1024 160 32 320" "3 2 1024 320 16 640" "1 2 1024 160 32 320" "1 2 1024 320 16 640" "3 1 1024 16 32 160" "1 1 1024 16 32 160"; do
    set -- $s; SV_BENCH_K=$1 SV_BENCH_S=$2 python3 "$R/tools/layer_bench.py" $3 $4 $5 $6; done; } 2>/dev/null | grep "of bf16" > "$OUT/${TAG}_layer_bench_odd.txt"
set -- "$TAG" "$PMC"
SV_BENCH_TABLE=1 python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline 2> "$OUT/${TAG}_bench_kernel_table.txt" > /dev/null
SV_BENCH_TABLE=1 python3 "$R/bench.py" --net wideresnet-28-10 --batch 256 --classes 100 --steps 10 --warmup 3 --no-cpu-baseline 2> "$OUT/${TAG}_cfg4_wrn28_10_kernel_table.txt" | tail -1 > "$OUT/${TAG}_cfg4_wrn28_10_bench.json"
rm -rf /tmp/prof_c
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c -o c -- python3 "$R/bench.py" --net wideresnet-28-10 --batch 256 --classes 100 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --wgrad-side 0 > /tmp/prof_c.log 2>&1
F=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/${TAG}_cfg4_wrn28_10_kernel_stats.csv"
python3 "$R/tools/kernel_stats_digest.py" "$F" 8 "rocprofv3 --kernel-trace --stats -- python3 bench.py --net wideresnet-28-10 --batch 256 --classes 100 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --wgrad-side 0   (BASELINE config 4; grouped schedule, eager, one stream)" > "$OUT/${TAG}_cfg4_wrn28_10_kernel_stats.txt"
# round 3: the odd layers of config 2 one by one, the step's timeline (idle gaps), config 5
bash "$R/tools/odd_layers.sh" > "$OUT/${TAG}_layer_bench_odd_cfg2.txt" 2>&1
(cd "$R" && python3 tools/step_timeline.py) > "$OUT/${TAG}_step_timeline.txt" 2>&1
python3 "$R/bench.py" --workload svhn --batch 1024 --steps 30 --warmup 5 2>/dev/null | tail -1 > "$OUT/${TAG}_svhn_bench.json"
if [ "$PMC" = "pmc" ]; then
  python3 "$R/tools/pmc_traffic.py" "$OUT/${TAG}_pmc_traffic.json" > /tmp/pmc.log 2>&1 || tail -5 /tmp/pmc.log
  (cd "$R" && bash tools/pmc_odd.sh) > "$OUT/${TAG}_pmc_odd.txt" 2>&1
fi
ls -la "$OUT"
