python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "thin_4x4_stride2_wgrad or thin_layers_wgrad" 2>&1 | tail -8
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "smooth" 2>&1 | tail -3
python tools/probes/svhn_layers.py 1024 10 2>&1 | grep "sv_wgrad\|^sum\|^wall"
python bench.py --workload svhn --batch 1024 --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-300
