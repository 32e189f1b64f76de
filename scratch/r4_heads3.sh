python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -2
for rep in 1 2 3; do
for lib in new old; do
  if [ $lib = old ]; then export HUGS_LIB_PATH=$PWD/scratch/libhugs_old_heads.so; else unset HUGS_LIB_PATH; fi
  python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $lib', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
done
done
for lib in new old; do
  if [ $lib = old ]; then export HUGS_LIB_PATH=$PWD/scratch/libhugs_old_heads.so; else unset HUGS_LIB_PATH; fi
  python bench.py --rays-per-gpu 128 --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128 rays $lib', d['ms_per_step'], d['value'])"
done
