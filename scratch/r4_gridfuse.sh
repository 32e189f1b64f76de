python -m pytest tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_fp16.py -q -x 2>&1 | grep -v "^E    \+" | tail -6
for rep in 1 2; do for v in 0 1; do
  HUGS_NF_GRID_FUSE=$v python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 GRID_FUSE=$v', d['ms_per_step'], d['value'])"
done; done
