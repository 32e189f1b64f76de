#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5cc; O=gpurun_out/r5cc; rm -f $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_vs_reference_model.py tests/test_gpu_determinism.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2 3; do for v in new old; do
  if [ $v = old ]; then export HUGS_LIB_PATH=$PWD/scratch/liboptold.so; else unset HUGS_LIB_PATH; fi
  python bench.py --rays-per-gpu 128 --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('128 rays $v', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
for v in new old new old; do
  if [ $v = old ]; then export HUGS_LIB_PATH=$PWD/scratch/liboptold.so; else unset HUGS_LIB_PATH; fi
  python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('cfg2 $v', d['ms_per_step'], d['loss_last'])" | tee -a $O/ab.txt
done
