#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5l; O=gpurun_out/r5l
python scratch/ipe_bench.py 2>&1 | grep -v amdgpu | tee $O/ipe.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_reference_properties.py tests/test_gpu_parity_tight.py tests/test_gpu_bench_config.py tests/test_gpu_vs_reference_model.py tests/test_gpu_eval_and_finetune.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 > $O/bench.json
python -c "import json;d=json.load(open('$O/bench.json'));print('cfg2', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
python bench.py --config ref360 --no-cpu-baseline --min-time 3 --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/ref360.json
python -c "import json;d=json.load(open('$O/ref360.json'));print('ref360', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
