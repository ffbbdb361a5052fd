#!/bin/bash
# Timing ablations of conv3x3w (diagnostic builds; results are numerically wrong by construction).
# usage: tools/w3_ablate.sh "NO_MFMA" "NO_W" ...   (each argument = one build, space-separated macro suffixes inside)
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
for v in "$@"; do
  D=""; for m in $v; do [ "$m" != "BASE" ] && D="$D -DSV_W3_$m"; done
  /opt/rocm/bin/hipcc $FLAGS $D -c conv3x3w.hip -o conv3x3w.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo "== $v"
  (cd ../.. && for shp in "512 160 32 160" "512 640 8 640"; do timeout 120 python tools/layer_bench.py $shp fwd 2>&1 | grep fwd; done)
done
