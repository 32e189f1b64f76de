#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5j; O=gpurun_out/r5j
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp256" 2>&1 | tail -3 | tee $O/pytest_mlp.txt
echo "--- 8 waves"; timeout 200 python scratch/mlpfuse_bench.py 2>&1 | grep -v amdgpu | tee $O/bench_fwd8.txt
echo "--- 4 waves"; HUGS_MLPFUSE_WAVES=4 timeout 200 python scratch/mlpfuse_bench.py 2>&1 | grep fused | tee $O/bench_fwd4.txt
for rep in 1 2; do
for w in 8 4; do
  echo "--- cfg2 waves=$w rep=$rep"; HUGS_MLPFUSE_WAVES=$w timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/ab_cfg2.txt
done; done
for w in 8 4; do
  echo "--- ref360 waves=$w"; HUGS_MLPFUSE_WAVES=$w timeout 400 python bench.py --config ref360 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" | tee -a $O/ab_ref360.txt
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_all.txt
