#!/bin/bash
# Build variant libraries of conv3x3x.hip for same-box A/Bs (run HERE, on the CPU box; build/ is shipped by gpurun, git-ignored):
#   bash tools/probes/x3_variants.sh "name:-DFLAG=.. -DFLAG=.." ...      -> build/ab/lib_<name>.so
# e.g.  "dmah0:-DSV_X3_DMAH=0" "stamp0c8:-DSV_X3_STAMP=1 -DSV_X3_CAP=8"
# then  gpurun -- 'bash tools/probes/x3_epi.sh'   (times every build/ab/lib_*.so)   or   'bash tools/probes/x3_stamps.sh 512'
R="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$R/build/ab"
cd "$R/shot_vae_amd/csrc"
FL="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function -Wno-unused-variable"
OBJS=""; for o in igemm halo tconv sconv pconv dconv thconv thwgrad s2wgrad hwgrad conv3x3 conv3x3w wgrad wgrad3x3 small runtime; do OBJS="$OBJS $o.o"; done
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc $FL $f -c conv3x3x.hip -o ../../build/ab/x3_$n.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/ab/x3_$n.o $OBJS -o ../../build/ab/lib_$n.so && rm ../../build/ab/x3_$n.o ) &
done
wait
ls -la ../../build/ab/
