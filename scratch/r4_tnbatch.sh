#!/bin/bash
mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_kernels.py -q -x -k "tn_batch or level_sample_bit" 2>&1 | tail -3
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_backward.py tests/test_gpu_vs_reference_model.py tests/test_gpu_data_parallel.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -5
for rep in 1 2; do
for b in 0 1 2; do
  HUGS_TN_BATCH=$b python bench.py --min-time 3 --no-cpu-baseline > gpurun_out/r4b/bench_tn$b.$rep.json 2> gpurun_out/r4b/bench_tn$b.$rep.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4b/bench_tn$b.$rep.json').read().strip().splitlines()[-1])
print('TN_BATCH=$b', d['ms_per_step'], d['value'], d['step_mfma_frac'], d['roofline']['avg_us'], [ (k['kernel'][:30], k['avg_us']) for k in d['instep_kernels']])
PY
done
done
