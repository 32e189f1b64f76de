#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5i
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp256" 2>&1 | tail -5 | tee gpurun_out/r5i/pytest.txt
echo "--- 8 waves"; timeout 200 python scratch/mlpfuse_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5i/bench_fwd8.txt
echo "--- 4 waves"; HUGS_MLPFUSE_WAVES=4 timeout 200 python scratch/mlpfuse_bench.py 2>&1 | grep fused | tee gpurun_out/r5i/bench_fwd4.txt
