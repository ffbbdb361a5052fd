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
# (every launch is matched against tools/pmc_traffic.py's ALL_KERNELS: the register-resident kernels of round 5 included)
run "fwd:dec4 (ConvT 128->64, 8x8 -> 16x16; tconvr)" SV_BENCH_T=1 -- 2048 128 8 64 fwd
run "wgrad:dec2 (ConvT 512->256, 2x2 -> 4x4)" SV_BENCH_T=1 -- 2048 512 2 256 wgrad
run "fwd:dec1 (ConvT 1024->512, 1x1 -> 2x2)" SV_BENCH_T=1 -- 2048 1024 1 512 fwd
run "dgrad:conv3x3_64x128_s2 (tconvr, activation-backward form)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 64 16 128 dgrad
run "dgrad:conv3x3_32x64_s2 (tconvr)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 32 32 64 dgrad
run "fwd:conv3x3_32x64_s2 (sconv)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 32 32 64 fwd
run "fwd:conv3x3_64x128_s2 (sconv)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 64 16 128 fwd
run "wgrad:conv3x3_32x64_s2 (s2wgrad)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 32 32 64 wgrad
run "wgrad:conv3x3_64x128_s2 (s2wgrad)" SV_BENCH_K=3 SV_BENCH_S=2 -- 2048 64 16 128 wgrad
run "fwd:conv1x1_16x32 (pconv)" SV_BENCH_K=1 SV_BENCH_S=1 -- 2048 16 32 32 fwd
run "fwd:conv1x1_32x64 s2 (pconv)" SV_BENCH_K=1 SV_BENCH_S=2 -- 2048 32 32 64 fwd
run "fwd:dec5 (ConvT 64->16, 16x16 -> 32x32; tconvx16)" SV_BENCH_T=1 -- 1024 64 16 16 fwd
run "dgrad:dec5 (dconv)" SV_BENCH_T=1 -- 1024 64 16 16 dgrad
run "dgrad:conv3x3_16x32_s1 (thconv)" SV_BENCH_NOPRO=1 -- 2048 16 32 32 dgrad
run "wgrad:stem (thwgrad)" SV_BENCH_NOPRO=1 -- 2048 16 32 16 wgrad
run "wgrad:conv3x3_16x32_s1 (thwgrad)" SV_BENCH_NOPRO=1 -- 2048 16 32 32 wgrad
run "fwd:stem" SV_BENCH_NOPRO=1 -- 2048 16 32 16 fwd
# round 6: the fused backward of the 32-channel body (bwd3x3f.hip) in the two forms the step runs
run "bwd:conv3x3_32x32_s1+bn (bwd3x3f, two-tensor form)" SV_BENCH_FUSED_BLOCKS=248 -- 2048 32 32 32 bwd2
run "bwd:conv3x3_32x32_s1+bn+skip (bwd3x3f, residual form)" SV_BENCH_FUSED_BLOCKS=248 -- 2048 32 32 32 bwd3
run "bwd:conv3x3_64x64_s1+bn (bwd3x3g, two-tensor form)" SV_BENCH_FUSED_BLOCKS=248 -- 2048 64 16 64 bwd2
run "bwd:conv3x3_64x64_s1+bn+skip (bwd3x3g, residual form)" SV_BENCH_FUSED_BLOCKS=248 -- 2048 64 16 64 bwd3
