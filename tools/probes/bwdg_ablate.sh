#!/bin/bash
# Timing ablations / variants of bwd3x3g_kernel (64 channels; ablated results are wrong by construction): scratch libraries
# build/ab/bwdg_<tag>.so, selected on the GPU box through SV_LIB_PATH.   build (here) | run (GPU box)
R="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$R/shot_vae_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
declare -A V=( [base]="" [nog]="-DSV_BWDG_ABL=1" [nod]="-DSV_BWDG_ABL=2" [noload]="-DSV_BWDG_ABL=4" [nostage]="-DSV_BWDG_ABL=8" [onlymma]="-DSV_BWDG_ABL=12" [donly]="-DSV_BWDG_ABL=13" [gonly]="-DSV_BWDG_ABL=14" [memonly]="-DSV_BWDG_ABL=3" [noepi]="-DSV_BWDG_ABL=16" [ks0]="-DSV_BWDG_KSHIFT=0" [dpp0]="-DSV_BWDG_DPP=0 -DSV_BWDG_PD=2" [hs1]="-DSV_BWDG_HSTG2=0" [w0]="-DSV_BWDG_WREG=0" [pd1]="-DSV_BWDG_PD=1" [pd3]="-DSV_BWDG_PD=3" [pd3donly]="-DSV_BWDG_PD=3 -DSV_BWDG_ABL=13" )
[ -n "$SV_BWDG_TAGS" ] || SV_BWDG_TAGS="base nog nod noload nostage onlymma donly gonly memonly noepi"
mkdir -p "$R/build/ab"
if [ "$1" = "build" ]; then
  make -s -j8 > /dev/null || exit 1
  OBJS=$(ls *.o | grep -v asan | grep -v '^bwd3x3g.o$' | tr '\n' ' ')
  for t in $SV_BWDG_TAGS; do
    ( /opt/rocm/bin/hipcc $FLAGS ${V[$t]} $SV_BWDG_EXTRA -c bwd3x3g.hip -o "$R/build/ab/bwdg_$t.o" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/build/ab/bwdg_$t.o" $OBJS -o "$R/build/ab/bwdg_$t.so" && echo "built $t" ) &
  done
  wait
  exit 0
fi
cd "$R"
for t in $SV_BWDG_TAGS; do
  printf "%-9s " $t
  SV_LIB_PATH="$R/build/ab/bwdg_$t.so" python tools/bwdf_bench.py ${SV_BWDG_ARGS:-512 16 4 248 64} 2>&1 | grep fused | awk '{printf "%s %s us   ", $1, $3} END {print ""}'
done
