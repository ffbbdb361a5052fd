#!/bin/bash
# A/B of compile-time variants of small.hip (-DSV_BNB_UNROLL2=0 ...): rebuild small.o, relink, run the step.  GPU box.
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc"
for rep in 1 2; do
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I$R/include $flags -c small.hip -o small.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm.o halo.o tconv.o sconv.o pconv.o dconv.o thconv.o thwgrad.o s2wgrad.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o -o ../libshotvae_hip.so
  echo -n "[$flags]  "; python3 $R/bench.py --no-extras --no-cpu-baseline --no-roofline --steps 60 2>/dev/null | grep -o "ms_per_step[^,]*"
done
done
