#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
timeout 900 python -m pytest tests/test_gpu_nerfacto_encodings.py tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_reference.py tests/test_gpu_nerfacto_fp16.py tests/test_gpu_parity_tight.py -x -q 2>&1 | tail -5 | tee gpurun_out/r5e/pytest.txt
for rep in 1 2; do for v in HUGS_HG_LDS_LEVELS=1 none; do
  envs=""; [ "$v" != none ] && envs=$v
  env $envs timeout 300 python bench.py --config cfg5 --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], [(k['kernel'][:28], k['ms_per_step']) for k in [d['roofline']]+d['instep_kernels'][:6]])" | tee -a gpurun_out/r5e/ab.txt
done; done
