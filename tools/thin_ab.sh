lb() { python3 tools/layer_bench.py "$@" 2>/dev/null | grep "of bf16" | awk '{printf "%s %s us | ", $(NF-9), $(NF-8)}'; }
for v in default A B C; do
  if [ $v = default ]; then unset SV_LIB_PATH; else export SV_LIB_PATH=$PWD/build/ab/lib_$v.so; fi
  echo "== $v"
  echo -n "stem: "; SV_BENCH_NOPRO=1 lb 2048 16 32 16 fwd; echo
  echo -n "3x3 16->32: "; lb 2048 16 32 32 fwd dgrad; echo
  echo -n "1x1 16->32: "; SV_BENCH_K=1 lb 2048 16 32 32 fwd dgrad; echo
  echo -n "3x3 s2 32->64: "; SV_BENCH_S=2 lb 2048 32 32 64 dgrad; echo
  echo -n "1x1 s2 32->64: "; SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 32 32 64 fwd dgrad; echo
  echo -n "1x1 s2 64->128: "; SV_BENCH_K=1 SV_BENCH_S=2 lb 2048 64 16 128 fwd dgrad; echo
  echo -n "dec5: "; SV_BENCH_T=1 lb 2048 64 16 16 fwd dgrad; echo
done
