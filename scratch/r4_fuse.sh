python -m pytest tests/test_gpu_kernels.py -q -x -k mlp256 2>&1 | tail -2
HUGS_MLP_FUSE_ROWS=100000000 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_step_graph.py tests/test_gpu_bench_config.py tests/test_gpu_psnr_equivalence.py -q -x 2>&1 | tail -3
for rep in 1 2; do for v in 0 32768; do
  HUGS_MLP_FUSE_ROWS=$v python bench.py --rays-per-gpu 128 --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('128 rays FUSE_ROWS=$v', d['ms_per_step'], d['value'])"
  HUGS_MLP_FUSE_ROWS=$v python bench.py --rays-per-gpu 256 --min-time 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256 rays FUSE_ROWS=$v', d['ms_per_step'], d['value'])"
done; done
