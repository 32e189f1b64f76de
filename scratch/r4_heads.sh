python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_vs_reference_model.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -4
python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
git stash -q; bash nerf-hugs_amd/csrc/build.sh > /dev/null 2>&1
python bench.py --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 BEFORE', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
python bench.py --config ref360 --min-time 3 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 BEFORE', d['ms_per_step'], d['value'], d['step_mfma_frac'])"
