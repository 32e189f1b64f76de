#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_data_parallel.py tests/test_gpu_train_step.py tests/test_gpu_determinism.py tests/test_gpu_eval_and_finetune.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5d/pytest.txt
for n in 128 256 512; do
  timeout 300 python bench.py --no-cpu-baseline --rays-per-gpu $n --min-time 3 --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['step_graph'])" | tee -a gpurun_out/r5d/small.txt
done
