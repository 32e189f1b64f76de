python scratch/ffuse_bench.py 2>&1 | grep -v amdgpu
python scratch/fbwd_bench.py 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_nerfacto.py -q -x -k "fused_field or cfg5" 2>&1 | grep -v "^E    " | tail -4
for rep in 1 2; do for v in "0 0" "1 1"; do set -- $v
  HUGS_NF_FIELD_FUSE=$1 HUGS_NF_FIELD_FUSE_BWD=$2 python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 FUSE=$1 BWD=$2', d['ms_per_step'], d['value'])"
done; done
