#!/bin/bash
# Timing ablations of wgrad3x3q_kernel (results wrong by construction): `build` here, `run` on the GPU box (kernel-only times)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
declare -A V=( [base]="" [nomma]="-DSV_WG3Q_NO_MMA" [noxform]="-DSV_WG3Q_NO_XFORM" [noload]="-DSV_WG3Q_NO_LOAD" [onlymma]="-DSV_WG3Q_NO_LOAD -DSV_WG3Q_NO_XFORM" [onlyload]="-DSV_WG3Q_NO_MMA -DSV_WG3Q_NO_XFORM" [halfreads]="-DSV_WG3Q_NO_LOAD -DSV_WG3Q_NO_XFORM -DSV_WG3Q_HALF_READS" [noreads]="-DSV_WG3Q_NO_LOAD -DSV_WG3Q_NO_XFORM -DSV_WG3Q_NO_READS" )
mkdir -p "$R/build/ab"
if [ "$1" = "build" ]; then
  make -s -j8 > /dev/null || exit 1
  OBJS=""; for o in igemm halo hwgrad conv3x3 conv3x3w conv3x3x wgrad small runtime; do OBJS="$OBJS $o.o"; done
  for t in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc $FLAGS ${V[$t]} -c wgrad3x3.hip -o "$R/build/ab/wg3q_$t.o" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/build/ab/wg3q_$t.o" $OBJS -o "$R/build/ab/wg3q_$t.so" && echo "built $t" ) &
  done
  wait
  exit 0
fi
cd "$R"
for t in ${SV_ABL:-base nomma noxform noload onlymma onlyload halfreads}; do
  for shape in "2048 64 16 64" "2048 128 8 128"; do
    printf "%-9s %-14s " $t "$shape"
    SV_LIB_PATH="$R/build/ab/wg3q_$t.so" SV_BENCH_ENABLE=131072 SV_BENCH_PERSISTENT_BLOCKS=256 tools/probes/ktrace.sh $shape wgrad | grep wgrad3x3q | awk '{print $1, "us"}'
  done
done
