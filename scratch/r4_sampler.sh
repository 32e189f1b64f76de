python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vs_reference_model.py -q -x 2>&1 | tail -3
python scratch/sampler_order_bench.py 2>&1 | tail -6
bash scratch/r4_switches.sh
