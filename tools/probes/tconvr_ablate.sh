#!/bin/bash
# Where the time of tconv.hip goes: rebuild tconv.o with one part of the loop body removed (SV_TCONVR_DBG bits: 1 no MFMA
# loop, 2 no output stores, 4 no next-image load / staging, 8 no statistics) or another schedule (-DSV_TCONVR_SKEW=0,
# -DSV_TCONVR_PD=1, ...), relink, time the layer.  Run on the GPU box: bash tools/probes/tconvr_ablate.sh "-DSV_TCONVR_DBG=1" ...
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$R/include $flags -c tconv.hip -o tconv.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o tconv.o sconv.o pconv.o dconv.o thconv.o thwgrad.o s2wgrad.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo -n "[$flags]  "; SV_BENCH_T=1 python3 $R/tools/layer_bench.py 2048 128 8 64 2>&1 | grep "of bf16" | head -1
done
