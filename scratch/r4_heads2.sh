for rep in 1 2; do
for lib in new old; do
  if [ $lib = old ]; then export HUGS_LIB_PATH=$PWD/scratch/libhugs_old_heads.so; else unset HUGS_LIB_PATH; fi
  python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $lib', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
  python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 $lib', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
done
done
