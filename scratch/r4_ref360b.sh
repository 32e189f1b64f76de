for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  HUGS_DW_AFTER_PROP=$1 HUGS_SIDE_LATE=$2 python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 DW_AFTER_PROP=$1 SIDE_LATE=$2', d['ms_per_step'], d['value'])"
done
for v in 0 1; do
  HUGS_DW_AFTER_PROP=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 DW_AFTER_PROP=$v', d['ms_per_step'], d['value'])"
done
