#!/bin/bash
# Variants of the 16x16 data-gradient kernel of tconv.hip (-DSV_TCONVX16_TP=2, -DSV_TCONVR_PD=1, ...): rebuild, relink, time.  GPU box.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$R/include $flags -c tconv.hip -o tconv.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o tconv.o sconv.o pconv.o dconv.o thconv.o thwgrad.o s2wgrad.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo -n "[$flags]  "; SV_BENCH_S=2 python3 $R/tools/layer_bench.py 2048 32 32 64 2>&1 | grep "dgrad" | head -1
done
