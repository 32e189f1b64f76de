python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_vs_reference_model.py tests/test_gpu_step_graph.py -q -x 2>&1 | tail -4
for rep in 1 2; do for v in 0 1; do
  HUGS_HEAD_FOLD=$v python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HEAD_FOLD=$v', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
done; done
for v in 0 1; do
  HUGS_HEAD_FOLD=$v python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 HEAD_FOLD=$v', d['ms_per_step'], d['value'])"
done
