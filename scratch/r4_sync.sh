python -m pytest tests/test_gpu_kernels.py -q -x -k "tn_batch" 2>&1 | tail -2
for rep in 1 2; do
for v in 0 1; do
  HUGS_TN_BATCH_SYNC=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TN_BATCH_SYNC=$v', d['ms_per_step'], d['value'], [(k['kernel'][:12], k['avg_us']) for k in d['instep_kernels']])"
done
done
HUGS_TN_BATCH_SYNC=1 bash scratch/pmc_run2.sh r04s > gpurun_out/pmc_r04s.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_r04s.json'))
for k,v in d.items():
  if 'batch' in k: print(k, v.get('FETCH_SIZE'), v.get('WRITE_SIZE'), v.get('avg_us_profiled'))
PY
