python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "smooth" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for k in fwd dgrad wgrad; do
  for s in "2048 64 16 64" "2048 128 8 128"; do
    python3 $R/tools/layer_bench.py $s $k 2>/dev/null | grep "of bf16"
    python3 $R/tools/pmc_sq.py $s $k 2>&1 | grep -v "^  SQ_[A-Z_]* *[0-9]*$"
  done
done > $R/gpurun_out/pmc_body_r06.txt 2>&1
cat $R/gpurun_out/pmc_body_r06.txt
