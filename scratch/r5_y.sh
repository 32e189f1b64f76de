#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5y; O=gpurun_out/r5y; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_step_graph.py tests/test_gpu_data_parallel.py tests/test_gpu_eval_and_finetune.py tests/test_gpu_bench_config.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for r in 128 256 128 256; do
  python bench.py --rays-per-gpu $r --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$r rays', d['ms_per_step'])" | tee -a $O/ab.txt
done
python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('cfg2', d['ms_per_step'])" | tee -a $O/ab.txt
