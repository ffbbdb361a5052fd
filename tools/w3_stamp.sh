#!/bin/bash
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS -DSV_W3_STAMP -c conv3x3w.hip -o conv3x3w.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
cd ../.. && python tools/w3_stamp.py 512 160 32 160 && python tools/w3_stamp.py 512 640 8 640
