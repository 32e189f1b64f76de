#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5f
timeout 600 python -m pytest tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_fp16.py tests/test_gpu_nerfacto_reference.py -x -q 2>&1 | tail -3 | tee gpurun_out/r5f/pytest.txt
for rep in 1 2; do for v in none HUGS_SIDE_LATE=1 HUGS_DW_AFTER_PROP=0; do
  envs=""; [ "$v" != none ] && envs=$v
  env $envs timeout 300 python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['fixed_batch']['ms_per_step'])" | tee -a gpurun_out/r5f/ab.txt
done; done
timeout 400 python bench.py --config cfg5 --no-cpu-baseline --min-time 4 > gpurun_out/r5f/cfg5.json 2>gpurun_out/r5f/cfg5.err; python -c "
import json; d=json.loads(open('gpurun_out/r5f/cfg5.json').read().strip().splitlines()[-1]); print('cfg5', d['value'], d['ms_per_step'], d.get('grid_input_gradient'), d.get('with_16bit_grid_input_gradient'), d['roofline']['kernel'][:30], d['roofline']['ms_per_step'])"
timeout 400 python bench.py --config cfg5 --dtype bf16 --no-cpu-baseline --min-time 3 > gpurun_out/r5f/cfg5_bf16.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/r5f/cfg5_bf16.json').read().strip().splitlines()[-1]); print('cfg5 bf16', d['value'], d['ms_per_step'], d.get('grid_input_gradient'), d.get('with_16bit_grid_input_gradient'))"
for D in zeros relu randn; do HUGS_LIB_PATH=$PWD/scratch/libhugs_trace.so DATA=$D timeout 120 python scratch/ntp_trace.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/r5f/nt_phase_trace.txt
