python -m pytest tests/test_gpu_nerfacto.py -q -x -k "fused_field or cfg5" 2>&1 | tail -3
for rep in 1 2 3; do
  python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 (c1 gradient behind the fused launch)', d['ms_per_step'], d['value'])"
  HUGS_NF_GRID_SIDE=0 python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 GRID_SIDE=0', d['ms_per_step'], d['value'])"
done
