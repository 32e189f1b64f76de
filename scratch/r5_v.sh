#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5v; O=gpurun_out/r5v
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_step_graph.py tests/test_gpu_eval_and_finetune.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2 3; do for v in 1 0; do
  HUGS_DENSITY_FORK=$v python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('fork=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done; done
for v in 1 0; do
  HUGS_DENSITY_FORK=$v python bench.py --config ref360 --no-cpu-baseline --min-time 3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('ref360 fork=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
