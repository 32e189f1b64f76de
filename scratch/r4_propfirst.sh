python -m pytest tests/test_gpu_nerfacto.py -q -x -k "fused_field or cfg5 or two_rank" 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 2; do
  HUGS_NF_PROP_FIRST=$v python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 PROP_FIRST=$v', d['ms_per_step'], d['value'])"
done; done
