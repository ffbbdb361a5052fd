#!/bin/bash
# Per-layer times of the "odd" layers of BASELINE config 2 (WRN-28-2 + decoder) at the grouped launch size (4 x 512 images):
# forward / data gradient / weight gradient through the C ABI.  Run on the GPU box: bash tools/odd_layers.sh
R="$(cd "$(dirname "$0")/.." && pwd)"
lb() { python3 "$R/tools/layer_bench.py" "$@" 2>/dev/null | grep "of bf16"; }
echo "# stem";            SV_BENCH_NOPRO=1 lb 2048 16 32 16
echo "# 3x3 16->32";      lb 2048 16 32 32
echo "# 1x1 16->32";      SV_BENCH_K=1 lb 2048 16 32 32
echo "# 3x3 s2 32->64";   SV_BENCH_S=2 lb 2048 32 32 64
echo "# 1x1 s2 32->64";   SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 32 32 64
echo "# 3x3 s2 64->128";  SV_BENCH_S=2 lb 2048 64 16 128
echo "# 1x1 s2 64->128";  SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 64 16 128
echo "# dec1 ConvT 1024->512 1x1->2x2";  SV_BENCH_T=1 lb 2048 1024 1 512
echo "# dec2 ConvT 512->256 2x2->4x4";   SV_BENCH_T=1 lb 2048 512 2 256
echo "# dec3 ConvT 256->128 4x4->8x8";   SV_BENCH_T=1 lb 2048 256 4 128
echo "# dec4 ConvT 128->64 8x8->16x16";  SV_BENCH_T=1 lb 2048 128 8 64
echo "# dec5 ConvT 64->3(16) 16x16->32x32"; SV_BENCH_T=1 lb 2048 64 16 16
