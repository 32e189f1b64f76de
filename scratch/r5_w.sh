#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5w; O=gpurun_out/r5w; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_step_graph.py tests/test_gpu_parity_tight.py tests/test_gpu_vs_reference_model.py tests/test_gpu_data_parallel.py tests/test_gpu_backward.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2 3; do for v in 1 0; do
  HUGS_INTERLEVEL_ON_PROP=$v HUGS_COMPOSITE_RAW=$v python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('new=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done; done
for v in 1 0; do
  HUGS_INTERLEVEL_ON_PROP=$v HUGS_COMPOSITE_RAW=$v python bench.py --rays-per-gpu 128 --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('128 rays new=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
  HUGS_INTERLEVEL_ON_PROP=$v HUGS_COMPOSITE_RAW=$v python bench.py --config ref360 --no-cpu-baseline --min-time 3 --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('ref360 new=$v', d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
