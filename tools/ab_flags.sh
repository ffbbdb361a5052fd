#!/bin/bash
# A/B of a compiler flag set on the same GPU box: layer bench + step bench with the committed build, then with EXTRA.
# usage: bash tools/ab_flags.sh "<extra hipcc flags>"
R="$(cd "$(dirname "$0")/.." && pwd)"
run() {
    python3 "$R/tools/layer_bench.py" 2>/dev/null | grep "of bf16"
    for s in "512 32 32 32" "512 64 16 64" "512 128 8 128"; do python3 "$R/tools/layer_bench.py" $s 2>/dev/null | grep "of bf16"; done
    python3 "$R/bench.py" --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
}
echo "== baseline build"; run
make -C "$R/shot_vae_amd/csrc" clean > /dev/null; make -C "$R/shot_vae_amd/csrc" -j8 EXTRA="$1" 2>&1 | grep -E "error" 
echo "== with $1"; run
