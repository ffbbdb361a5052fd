#!/bin/bash
# Where the time of swgrad.hip goes: rebuild with parts compiled out (SV_SWG_DBG bits: 1 no final atomics, 2 no MFMA loop, 4 no
# staging of the next image), relink, time the layer.  GPU box.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc"
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$R/include $flags -c swgrad.hip -o swgrad.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o tconv.o sconv.o cconv.o swgrad.o pconv.o dconv.o thconv.o thwgrad.o s2wgrad.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo -n "[$flags]  "; SV_BENCH_S=2 python3 $R/tools/layer_bench.py 2048 64 16 128 2>&1 | grep "wgrad" | head -1
done
