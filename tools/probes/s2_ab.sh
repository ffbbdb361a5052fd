#!/bin/bash
# dispatch alternatives of the stride-2 1x1 shortcut layers (forward / data gradient / weight gradient), config 2 and config 4 sizes
R="$(cd "$(dirname "$0")/../.." && pwd)"
lb() { python3 "$R/tools/layer_bench.py" "$@" 2>/dev/null | grep "of bf16" | cut -c1-110; }
for m in 0 512 768 16896; do
echo "## disable=$m"
SV_BENCH_DISABLE=$m SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 32 32 64
SV_BENCH_DISABLE=$m SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 64 16 128
SV_BENCH_DISABLE=$m SV_BENCH_K=1 SV_BENCH_S=2 lb 1024 160 32 320
done
