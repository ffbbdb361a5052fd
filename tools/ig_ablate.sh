#!/bin/bash
# Timing ablations of the generic gather-GEMM (igemm.hip; diagnostic builds, numerically wrong by construction).
# usage: tools/ig_ablate.sh "BASE" "NO_MFMA" "NO_XF NO_GA" ...   (each argument = one build of macro suffixes)
# shapes: SV_IG_SHAPES="B Cin H N k s what; ..." (default: the WRN-28-10 stride-2 3x3 layer, forward and data gradient)
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
OBJS="halo.o hwgrad.o conv3x3.o conv3x3w.o conv3x3x.o wgrad.o wgrad3x3.o small.o runtime.o"
SHAPES=${SV_IG_SHAPES:-"1024 160 32 320 3 2 fwd;1024 160 32 320 3 2 dgrad"}
for v in "$@"; do
  D=""; for m in $v; do [ "$m" != "BASE" ] && D="$D -DSV_IG_$m"; done
  /opt/rocm/bin/hipcc $FLAGS $D $SV_IG_EXTRA -c igemm.hip -o igemm_abl.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC igemm_abl.o $OBJS -o ../libshotvae_hip.so
  echo "== $v"
  (cd ../.. && IFS=';' && for shp in $SHAPES; do IFS=' ' read -r B Ci H N K S W <<< "$shp"; SV_BENCH_K=$K SV_BENCH_S=$S timeout 120 python tools/layer_bench.py $B $Ci $H $N $W 2>&1 | grep " us "; done)
done
