for l in scratch/libhugs_old.so nerf-hugs_amd/csrc/libhugs_hip.so; do echo $l; HUGS_LIB_PATH=$PWD/$l python scratch/tn_cfg5_bench.py 2>&1 | grep -v amdgpu; done
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_nerfacto_fp16.py -q -x -k "tn or gemm" 2>&1 | tail -2
for rep in 1 2; do for l in scratch/libhugs_old.so nerf-hugs_amd/csrc/libhugs_hip.so; do
  HUGS_LIB_PATH=$PWD/$l python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 $l', d['ms_per_step'], d['value'])"
done; done
