#!/bin/bash
# round 5: NT epilogue trims (main) vs cached output stores, on the same box; numerics of the main build first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py tests/test_gpu_step_graph.py tests/test_gpu_nerfacto_fp16.py -x -q 2>&1 | tail -3 | tee gpurun_out/r5b/pytest.txt
for rep in 1 2; do for v in plainst main; do
  if [ $v = main ]; then L=$PWD/nerf-hugs_amd/csrc/libhugs_hip.so; else L=$PWD/scratch/lib$v.so; fi
  HUGS_LIB_PATH=$L timeout 300 python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['fixed_batch']['ms_per_step'], d['roofline']['avg_us'], [k['avg_us'] for k in d['instep_kernels']])" | tee -a gpurun_out/r5b/ab.txt
done; done
