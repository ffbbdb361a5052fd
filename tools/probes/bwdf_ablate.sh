#!/bin/bash
# Timing ablations / variants of bwd3x3f_kernel (ablated results are wrong by construction): built HERE into scratch libraries
# build/ab/bwdf_<tag>.so, selected on the GPU box through SV_LIB_PATH.
#   tools/probes/bwdf_ablate.sh build      (in the build container: hipcc cross-compiles)
#   tools/probes/bwdf_ablate.sh run        (on the GPU box)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
declare -A V=( [base]="" [nog]="-DSV_BWDF_ABL=1" [nod]="-DSV_BWDF_ABL=2" [noload]="-DSV_BWDF_ABL=4" [nostage]="-DSV_BWDF_ABL=8" [onlymma]="-DSV_BWDF_ABL=12" [noepi]="-DSV_BWDF_ABL=16" [donly]="-DSV_BWDF_ABL=13" [gonly]="-DSV_BWDF_ABL=14" [memonly]="-DSV_BWDF_ABL=3" [d0g0]="-DSV_BWDF_DFIRST=0 -DSV_BWDF_GFIRST=0" [d0g1]="-DSV_BWDF_DFIRST=0 -DSV_BWDF_GFIRST=1" [d1g1]="-DSV_BWDF_DFIRST=1 -DSV_BWDF_GFIRST=1" [frags2]="-DSV_BWDF_FRAGS=2" [frags3]="-DSV_BWDF_FRAGS=3" [wregs1]="-DSV_BWDF_WREGS=1" [g16]="-DSV_BWDF_G32=0 -DSV_BWDF_WREGS=2" )
[ -n "$SV_BWDF_TAGS" ] || SV_BWDF_TAGS="base nog nod noload nostage onlymma noepi"
mkdir -p "$R/build/ab"
if [ "$1" = "build" ]; then
  make -s -j8 > /dev/null || exit 1
  OBJS=$(ls *.o | grep -v asan | grep -v '^bwd3x3f.o$' | tr '\n' ' ')
  for t in $SV_BWDF_TAGS; do
    ( /opt/rocm/bin/hipcc $FLAGS ${V[$t]} $SV_BWDF_EXTRA -c bwd3x3f.hip -o "$R/build/ab/bwdf_$t.o" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/build/ab/bwdf_$t.o" $OBJS -o "$R/build/ab/bwdf_$t.so" && echo "built $t" ) &
  done
  wait
  exit 0
fi
cd "$R"
for t in $SV_BWDF_TAGS; do
  printf "%-9s " $t
  SV_LIB_PATH="$R/build/ab/bwdf_$t.so" python tools/bwdf_bench.py ${SV_BWDF_ARGS:-512 32 4 0} 2>&1 | grep fused | awk '{printf "%s %s us   ", $1, $3} END {print ""}'
done
