# decoder layers (grouped launch size of config 2: 4 x 512 images forward, 2 x 512 backward) under the dispatch knobs of igemm.hip
lb() { python3 tools/layer_bench.py "$@" 2>/dev/null | grep "of bf16" | awk '{printf "%s %s | ", $6, $7}'; }
for cfg in "default:" "no256:SV_BENCH_WIDE_MIN_BLOCKS=1000000" "nodma:SV_BENCH_DISABLE=16384" "nobig:SV_BENCH_DISABLE=2048" "min512:SV_BENCH_WIDE_MIN_BLOCKS=512"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name"
  for L in "2048 1024 1 512" "2048 512 2 256" "2048 256 4 128"; do
    echo -n "fwd($L): "; env $envs SV_BENCH_T=1 SV_BENCH_NOPRO=1 bash -c "$(declare -f lb); lb $L fwd"; echo
  done
  for L in "1024 1024 1 512" "1024 512 2 256" "1024 256 4 128" "1024 128 8 64"; do
    echo -n "bwd($L): "; env $envs SV_BENCH_T=1 bash -c "$(declare -f lb); lb $L dgrad"; echo
  done
done
