#!/bin/bash
# Repeats the config-4 bench N times and counts failures (non-finite loss / crash).
# usage: cfg4_repeat.sh N [env assignments]   (BENCH_FLAGS="--overlap 0" adds bench flags)
N=$1; shift
ok=0; bad=0
for i in $(seq 1 $N); do
  out=$(env "$@" python3 bench.py --net wideresnet-28-10 --batch 256 --classes 100 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline $BENCH_FLAGS 2>&1 | tail -2)
  if echo "$out" | grep -q "ms_per_step"; then ok=$((ok+1)); else bad=$((bad+1)); echo "FAIL $i: $(echo "$out" | tail -1 | cut -c1-200)"; fi
done
echo "[$* $BENCH_FLAGS] ok $ok bad $bad"
