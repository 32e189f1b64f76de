python -m pytest tests/test_gpu_kernels.py -q -x -k "ring or gemm_nt or bits" 2>&1 | tail -2
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_parity_tight.py tests/test_gpu_step_graph.py tests/test_gpu_bench_config.py -q -x 2>&1 | tail -3
for rep in 1 2 3; do for lib in scratch/libhugs_old.so nerf-hugs_amd/csrc/libhugs_hip.so; do
  HUGS_LIB_PATH=$PWD/$lib python bench.py --min-time 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $lib', d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'))"
done; done
for lib in scratch/libhugs_old.so nerf-hugs_amd/csrc/libhugs_hip.so; do
  HUGS_LIB_PATH=$PWD/$lib python bench.py --config ref360 --min-time 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ref360 $lib', d['ms_per_step'], d['value'])"
done
