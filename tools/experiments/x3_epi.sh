#!/bin/bash
# conv3x3x variants (build/ab/lib_*.so from tools/probes/x3_variants.sh) against the shipped library, same box; wide-kernel tests first.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
if [ "$1" = "test" ]; then timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_wide_gpu.py -q -x -k "one_wave or wide or conv3x3x or x3" 2>&1 | tail -4; fi
for rep in 1 2; do
for lib in shot_vae_amd/libshotvae_hip.so $(ls build/ab/lib_*.so); do
  echo -n "$lib  "
  for shape in "512 160 32 160" "512 320 16 320" "512 640 8 640"; do
    SV_LIB_PATH=$PWD/$lib timeout 300 python tools/layer_bench.py $shape 2>&1 | grep "of bf16" | grep -v wgrad | awk '{printf "%s %s us   ", $5, $6}'
  done; echo
done
done
