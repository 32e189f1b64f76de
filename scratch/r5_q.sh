#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5q; O=gpurun_out/r5q
timeout 900 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_train_step.py tests/test_gpu_vs_reference_model.py tests/test_gpu_data_parallel.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 600 python scratch/transient_perf.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/transient_perf.txt
HUGS_STEP_GRAPH=0 timeout 600 python scratch/transient_perf.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/transient_perf_eager.txt
