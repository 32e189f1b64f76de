#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5n; O=gpurun_out/r5n
echo "--- new"; python scratch/hg_lm.py 2>&1 | grep -v amdgpu | tee $O/hg_new.txt
echo "--- old"; HUGS_LIB_PATH=$PWD/scratch/libhgold.so python scratch/hg_lm.py 2>&1 | grep -v amdgpu | tee $O/hg_old.txt
timeout 900 python -m pytest tests/test_gpu_nerfacto.py tests/test_gpu_nerfacto_reference.py tests/test_gpu_vs_reference_model.py tests/test_gpu_step_graph.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for v in new old; do
  if [ $v = old ]; then export HUGS_LIB_PATH=$PWD/scratch/libhgold.so; else unset HUGS_LIB_PATH; fi
  python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 2>/dev/null | tail -1 > $O/cfg5_$v.json
  python -c "import json;d=json.load(open('$O/cfg5_$v.json'));print('cfg5 $v', d['ms_per_step'], d['value'])"
done
