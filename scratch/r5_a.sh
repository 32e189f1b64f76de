#!/bin/bash
# round 5, first GPU call: full GPU suite, bench (fresh-batch pool), K-rotation A/B, PMC traffic of the rotated kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5a/pytest.log
tail -5 gpurun_out/r5a/pytest.log
timeout 600 python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err; tail -c 3000 gpurun_out/r5a/bench.json
for rep in 1 2; do for v in krot0 main; do
  if [ $v = main ]; then L=$PWD/nerf-hugs_amd/csrc/libhugs_hip.so; else L=$PWD/scratch/lib$v.so; fi
  HUGS_LIB_PATH=$L timeout 300 python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['fixed_batch']['ms_per_step'], d['roofline']['avg_us'], [k['avg_us'] for k in d['instep_kernels']])" | tee -a gpurun_out/r5a/ab.txt
done; done
bash scratch/pmc_run2.sh r05 > gpurun_out/r5a/pmc.log 2>&1; tail -3 gpurun_out/r5a/pmc.log
