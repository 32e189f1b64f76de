#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5h
HUGS_LIB_PATH=$PWD/scratch/libhalf.so timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_backward.py -x -q 2>&1 | tail -3 | tee gpurun_out/r5h/pytest_half.txt
for rep in 1 2 3; do for v in half main; do
  if [ $v = main ]; then L=$PWD/nerf-hugs_amd/csrc/libhugs_hip.so; else L=$PWD/scratch/lib$v.so; fi
  HUGS_LIB_PATH=$L timeout 300 python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['fixed_batch']['ms_per_step'], d['roofline']['avg_us'], [k['avg_us'] for k in d['instep_kernels']])" | tee -a gpurun_out/r5h/ab.txt
done; done
