#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5o; O=gpurun_out/r5o
timeout 900 python -m pytest tests/test_gpu_step_graph.py -x -q 2>&1 | tail -15 | tee $O/pytest_graph.txt
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_data_parallel.py tests/test_gpu_vs_reference_model.py -x -q 2>&1 | tail -5 | tee $O/pytest2.txt
timeout 600 python scratch/transient_perf.py 2>&1 | grep -v amdgpu | tail -12 | tee $O/transient_perf.txt
