#!/bin/bash
# Diagnostic build of tconv.hip with s_memtime stamps (SV_TCONVR_DBG = 32) + tools/probes/tconvr_stamps.py.  GPU box.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc"
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$R/include -DSV_TCONVR_DBG=32 $f -c tconv.hip -o tconv.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o tconv.o sconv.o pconv.o dconv.o thconv.o thwgrad.o s2wgrad.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo "== $f"; python3 $R/tools/probes/tconvr_stamps.py 2048
done
