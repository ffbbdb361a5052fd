python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "register_resident_last_decoder or first_svhn" 2>&1 | tail -5
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "smooth" 2>&1 | tail -3
python tools/probes/svhn_layers.py 1024 10 2>&1 | grep "sv_igemm\|^sum\|^wall"
python bench.py --workload svhn --batch 1024 --steps 30 --warmup 5 2>/dev/null | tail -1 | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-200
