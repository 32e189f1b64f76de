#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5s; O=gpurun_out/r5s
echo "--- 8 waves"; timeout 300 python scratch/ffuse_bench.py 2>&1 | grep -v amdgpu | tee $O/fwd8.txt
echo "--- 4 waves"; HUGS_FF_WAVES=4 timeout 300 python scratch/ffuse_bench.py 2>&1 | grep -v amdgpu | tee $O/fwd4.txt
timeout 900 python -m pytest tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_reference.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for w in 8 4 8 4; do
  HUGS_FF_WAVES=$w python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('cfg5 waves $w', d['ms_per_step'], d['value'])" | tee -a $O/cfg5_ab.txt
done
