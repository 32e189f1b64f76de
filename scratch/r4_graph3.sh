python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_data_parallel.py -q -x 2>&1 | tail -15
python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | cut -c1-700
python bench.py --min-time 3 --no-cpu-baseline --rays-per-gpu 128 2>/dev/null | cut -c1-600
