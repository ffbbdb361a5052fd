#!/bin/bash
# A/B builds of ONE kernel file with macro sets, each followed by the per-launch table of the whole step (config 2 and,
# with SV_AB_C4=1, config 4).  Diagnostic; the shipped library is restored by `make`.
#   usage: tools/ab.sh igemm.hip "tags-regex" "-DA" "-DB -DC" ...
cd "$(dirname "$0")/../shot_vae_amd/csrc" || exit 1
FILE=$1; TAGS=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
ALL="igemm halo hwgrad conv3x3 conv3x3w conv3x3x wgrad wgrad3x3 small runtime"
OBJS=""; for o in $ALL; do [ "$o.hip" != "$FILE" ] && OBJS="$OBJS $o.o"; done
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $v -c $FILE -o ab_tmp.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab_tmp.o $OBJS -o ../libshotvae_hip.so
  echo "== $FILE $v"
  (cd ../.. && SV_BENCH_TABLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> /tmp/ab.table | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2:', d['ms_per_step'], 'ms', d['value'])"; grep -E "$TAGS" /tmp/ab.table | awk '{printf "   %-30s %8.1f us\n",$2,$6}'
   if [ -n "$SV_AB_C4" ]; then SV_BENCH_TABLE=1 python bench.py --net wideresnet-28-10 --classes 100 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline 2> /tmp/ab4.table | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4:', d['ms_per_step'], 'ms', d['value'])"; grep -E "$TAGS" /tmp/ab4.table | awk '{printf "   %-30s %8.1f us\n",$2,$6}'; fi)
done
rm -f ab_tmp.o
