#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "mlp256" 2>&1 | tail -5 | tee gpurun_out/r5c/pytest.txt
echo "--- chain3 (register-resident)"; timeout 200 python scratch/mlpfuse_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5c/bench_chain3.txt
timeout 200 python scratch/mlpfuse_bwd_bench.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5c/bench_chain3_bwd.txt
timeout 900 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_bench_config.py tests/test_gpu_step_graph.py tests/test_gpu_data_parallel.py tests/test_gpu_vs_reference_model.py tests/test_gpu_psnr_equivalence.py -x -q 2>&1 | tail -5 | tee -a gpurun_out/r5c/pytest.txt
for rep in 1 2; do for v in HUGS_MLPFUSE_CHAIN3=0 none; do
  envs=""; [ "$v" != none ] && envs=$v
  env $envs timeout 300 python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['fixed_batch']['ms_per_step'])" | tee -a gpurun_out/r5c/ab.txt
done; done
for v in HUGS_MLPFUSE_CHAIN3=0 none; do
  envs=""; [ "$v" != none ] && envs=$v
  env $envs timeout 400 python bench.py --config ref360 --no-cpu-baseline --min-time 4 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref360 $v', d['value'], d['ms_per_step'], d['step_mfma_frac'])" | tee -a gpurun_out/r5c/ab.txt
done
