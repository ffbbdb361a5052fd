mkdir -p gpurun_out/r03p
python -m pytest tests/test_kernels_gpu.py -q -k "halo or convT or thin" -p no:cacheprovider 2>&1 | tail -3
lb() { python3 tools/layer_bench.py "$@" 2>/dev/null | grep "of bf16" | sed "s/  */ /g"; }
for v in base new; do
  if [ $v = new ]; then unset SV_LIB_PATH; else export SV_LIB_PATH=$PWD/build/ab/lib_base.so; fi
  echo "== $v"
  echo -n "dec4 fwd: "; SV_BENCH_T=1 lb 2048 128 8 64 fwd
  echo -n "3x3 s2 64->128 dgrad: "; SV_BENCH_K=3 SV_BENCH_S=2 lb 2048 64 16 128 dgrad
  echo -n "cfg4 160->16 dgrad: "; lb 1024 16 32 160 dgrad
  SV_BENCH_TABLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /tmp/t.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2:', d['ms_per_step'])"
  grep -E "fwd:dec4|dgrad:conv3x3_64x128_s2" /tmp/t.txt
done
