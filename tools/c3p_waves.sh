#!/bin/bash
# A/B of conv3x3p at 2 vs 3 waves per SIMD (diagnostic rebuild of conv3x3.o on the GPU box)
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
for w in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DSV_C3P_WAVES=$w -c conv3x3.hip -o conv3x3.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo "== SV_C3P_WAVES=$w"
  (cd ../.. && for shp in "512 32 32 32" "512 64 16 64"; do timeout 120 python tools/layer_bench.py $shp fwd dgrad 2>&1 | grep us; done)
done
