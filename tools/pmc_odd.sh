#!/bin/bash
# PMC traffic + SQ counters of the odd layers the round-2 verdict names, at the grouped launch size of BASELINE config 2
# (4 x 512 images).  Run on the GPU box from the repo root: bash tools/pmc_odd.sh > gpurun_out/pmc_odd.txt
R="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
run() {   # name, env..., -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "### $name"
  env "${envs[@]}" python3 "$R/tools/layer_bench.py" "$@" 2>/dev/null | grep "of bf16"
  env "${envs[@]}" python3 "$R/tools/pmc_layer.py" "$@" 2>&1 | tail -1
  env "${envs[@]}" python3 "$R/tools/pmc_sq.py" "$@" 2>&1 | grep -v "^  SQ_[A-Z_]* *[0-9]*$"
}
run "fwd:dec4 (ConvT 128->64, 8x8 -> 16x16)" SV_BENCH_T=1 -- 2048 128 8 64 fwd
run "wgrad:dec2 (ConvT 512->256, 2x2 -> 4x4)" SV_BENCH_T=1 -- 2048 512 2 256 wgrad
run "fwd:dec1 (ConvT 1024->512, 1x1 -> 2x2)" SV_BENCH_T=1 -- 2048 1024 1 512 fwd
run "dgrad:conv3x3_64x128_s2" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 64 16 128 dgrad
run "fwd:stem" SV_BENCH_NOPRO=1 -- 2048 16 32 16 fwd
