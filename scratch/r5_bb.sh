#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5bb; O=gpurun_out/r5bb; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_train_step.py tests/test_gpu_vs_reference_model.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2 3; do for v in 1 0; do
  HUGS_OPT_STATS_FUSED=$v python bench.py --rays-per-gpu 128 --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('128 rays fused=$v', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
for v in 1 0; do
  HUGS_OPT_STATS_FUSED=$v python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('cfg2 fused=$v', d['ms_per_step'], d['loss_last'])" | tee -a $O/ab.txt
done
