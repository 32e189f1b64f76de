for rep in 1 2; do
for v in 0 1; do
  HUGS_SIDE_LATE=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SIDE_LATE=$v', d['ms_per_step'], d['value'], d['step_graph'])"
done
done
