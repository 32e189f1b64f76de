for rep in 1 2 3; do for v in 0 1; do
  HUGS_HEAD_FOLD=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 HEAD_FOLD=$v', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
done; done
python bench.py --rays-per-gpu 128 --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128 rays', d['ms_per_step'], d['value'])"
