#!/bin/bash
# A/B builds of ONE kernel file with macro sets (the tunables that are left in the sources: SV_HALOP_OCC*, SV_C3P_WAVES,
# SV_X3_EPD, SV_W3_EPD, SV_WG3_LD?_PAD, or any experiment of the moment), each followed by the per-launch table of the whole
# step (config 2 and, with SV_AB_C4=1, config 4).
#   usage: tools/ab.sh igemm.hip "tags-regex" "-DA=1" "-DB=2 -DC=3" ...
# Every variant is built into a SCRATCH library build/ab/lib_<n>.so from scratch objects and selected through SV_LIB_PATH
# (shot_vae_amd/_lib.py): the shipped shot_vae_amd/libshotvae_hip.so and its objects are never touched, compiler errors are
# shown, and a variant that does not build is reported as FAILED and not timed.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R/shot_vae_amd/csrc" || exit 1
FILE=$1; TAGS=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
ALL="igemm halo hwgrad conv3x3 conv3x3w conv3x3x wgrad wgrad3x3 small runtime"
make -s -j8 > /dev/null || { echo "FAILED: the shipped library does not build"; exit 1; }
OBJS=""; for o in $ALL; do [ "$o.hip" != "$FILE" ] && OBJS="$OBJS $o.o"; done
mkdir -p "$R/build/ab"
n=0
for v in "$@"; do
  n=$((n + 1))
  LIBV="$R/build/ab/lib_$n.so"
  echo "== $FILE $v"
  if ! /opt/rocm/bin/hipcc $FLAGS $v -c "$FILE" -o "$R/build/ab/ab_$n.o" || ! /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$R/build/ab/ab_$n.o" $OBJS -o "$LIBV"; then
    echo "   FAILED to build: not timed"
    continue
  fi
  (cd "$R" && export SV_LIB_PATH="$LIBV"
   SV_BENCH_TABLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /tmp/ab.table | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2:', d['ms_per_step'], 'ms', d['value'])"; grep -E "$TAGS" /tmp/ab.table | awk '{printf "   %-30s %8.1f us\n",$2,$6}'
   if [ -n "$SV_AB_C4" ]; then SV_BENCH_TABLE=1 python bench.py --net wideresnet-28-10 --classes 100 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline 2> /tmp/ab4.table | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4:', d['ms_per_step'], 'ms', d['value'])"; grep -E "$TAGS" /tmp/ab4.table | awk '{printf "   %-30s %8.1f us\n",$2,$6}'; fi)
done
