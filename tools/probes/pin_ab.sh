mkdir -p gpurun_out/r03q
python -m pytest tests/test_kernels_gpu.py tests/test_wide_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -3
for v in base new; do
  if [ $v = new ]; then unset SV_LIB_PATH; else export SV_LIB_PATH=$PWD/build/ab/lib_base.so; fi
  echo "== $v"
  SV_BENCH_TABLE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> gpurun_out/r03q/table_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2:', d['ms_per_step'])"
  head -32 gpurun_out/r03q/table_$v.txt | awk '{printf "   %-30s %8.1f us\n",$2,$6}'
  python tools/layer_bench.py 2>/dev/null | grep "of bf16"
  SV_BENCH_TABLE=1 python bench.py --net wideresnet-28-10 --classes 100 --batch 256 --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/r03q/table4_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4:', d['ms_per_step'])"
done
