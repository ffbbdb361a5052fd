#!/bin/bash
# Builds conv3x3x.hip with extra flags, then repeats the config-4 bench.  usage: x3_variant_repeat.sh "<flags>" N
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $1 -c conv3x3x.hip -o conv3x3x.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
cd ../.. && bash tools/cfg4_repeat.sh $2 SV_CONV3X3X=1 SV_VARIANT="$1"
