python -m pytest tests/test_fused_bwd_gpu.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_timed_path_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2 3; do for e in 0 1; do echo "fold=$e"; python bench.py --steps 30 --warmup 8 --no-extras --no-cpu-baseline --no-roofline --fold-bn-bwd $e 2>/dev/null | tail -1 | cut -c1-175; done; done
