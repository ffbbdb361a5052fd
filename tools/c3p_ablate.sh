#!/bin/bash
# Timing ablations of conv3x3p (diagnostic builds; results are numerically wrong by construction).
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
for v in "$@"; do
  D=""; for m in $v; do [ "$m" != "BASE" ] && D="$D -DSV_C3P_$m"; done
  /opt/rocm/bin/hipcc $FLAGS $D -c conv3x3.hip -o conv3x3.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo "== $v"
  (cd ../.. && timeout 120 python tools/layer_bench.py 512 32 32 32 fwd dgrad 2>&1 | grep us)
done
