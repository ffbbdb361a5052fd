#!/bin/bash
# Timing ablations of wgrad3x3m_kernel (results wrong by construction): built HERE into scratch libraries build/ab/wg3m_<tag>.so
#   tools/probes/wg3m_ablate.sh build      (in the build container: hipcc cross-compiles)
#   tools/probes/wg3m_ablate.sh run        (on the GPU box)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
declare -A V=( [base]="" [nomma]="-DSV_WG3M_NO_MMA" [noxform]="-DSV_WG3M_NO_XFORM" [noload]="-DSV_WG3M_NO_LOAD" [nohst]="-DSV_WG3M_NO_HST" [onlyload]="-DSV_WG3M_NO_MMA -DSV_WG3M_NO_HST" [onlymma]="-DSV_WG3M_NO_LOAD -DSV_WG3M_NO_HST" )
mkdir -p "$R/build/ab"
if [ "$1" = "build" ]; then
  make -s -j8 > /dev/null || exit 1
  OBJS=""; for o in igemm halo hwgrad conv3x3 conv3x3w conv3x3x wgrad small runtime; do OBJS="$OBJS $o.o"; done
  for t in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc $FLAGS ${V[$t]} $SV_WG3M_EXTRA -c wgrad3x3.hip -o "$R/build/ab/wg3m_$t.o" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/build/ab/wg3m_$t.o" $OBJS -o "$R/build/ab/wg3m_$t.so" && echo "built $t" ) &
  done
  wait
  exit 0
fi
cd "$R"
for t in base nomma noxform noload nohst onlyload onlymma; do
  for shape in "2048 32 32 32" "2048 64 16 64" "2048 128 8 128"; do
    for pb in 512 256; do
      printf "%-9s budget=%d  " $t $pb
      SV_LIB_PATH="$R/build/ab/wg3m_$t.so" SV_BENCH_PERSISTENT_BLOCKS=$pb python tools/layer_bench.py $shape wgrad 2>&1 | grep wgrad | awk '{print $1,$2,$3,$4,$6,"us"}'
    done
  done
done
