python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "smooth" 2>&1 | tail -3
python bench.py --workload svhn --batch 1024 --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-700
python bench.py --workload svhn --batch 1024 --steps 30 --warmup 5 --graph 0 2>/dev/null | tail -1 | cut -c1-200
