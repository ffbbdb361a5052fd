#!/bin/bash
# A/B of the side-stream fork of the paired backward: start signal (flag) against event forks; checks that no wait timed out
R="$(cd "$(dirname "$0")/../.." && pwd)"
for ff in 0 1 0 1; do
  python3 "$R/bench.py" --no-cpu-baseline --no-extras --no-roofline --steps 40 --flag-fork $ff 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('flag_fork=$ff', d['ms_per_step'], d['value'])"
done
