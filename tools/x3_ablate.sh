#!/bin/bash
# Timing experiments on conv3x3x (diagnostic builds).  usage: tools/x3_ablate.sh "<flags>" "<flags>" ...
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
for v in "$@"; do
  [ "$v" = "BASE" ] && D="" || D="$v"
  /opt/rocm/bin/hipcc $FLAGS $D -c conv3x3x.hip -o conv3x3x.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo "== $v"
  (cd ../.. && for shp in "512 160 32 160" "512 320 16 320" "512 640 8 640"; do timeout 120 python tools/layer_bench.py $shp fwd 2>&1 | grep fwd; done)
done
