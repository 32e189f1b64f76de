python -m pytest tests/test_gpu_kernels.py -q -x -k "ring_kernels or gemm_nt_tn" 2>&1 | tail -3
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_bench_config.py tests/test_gpu_nerfacto_fp16.py -q -x 2>&1 | tail -2
for v in 1 0; do
  HUGS_HEAD_FOLD=$v python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 HEAD_FOLD=$v', d['ms_per_step'], d['value'])"
  HUGS_HEAD_FOLD=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 HEAD_FOLD=$v', d['ms_per_step'], d['value'])"
  HUGS_HEAD_FOLD=$v python bench.py --config cfg3 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 HEAD_FOLD=$v', d['ms_per_step'], d['value'])"
done
