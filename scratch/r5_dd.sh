#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5dd; O=gpurun_out/r5dd; rm -f $O/ab3.txt
for rep in 1 2 3; do for lanes in "1,3,4" "0,1,3,4,5"; do
  HUGS_STEP_GRAPH_LANES="$lanes" timeout 300 python bench.py --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('cfg2 lanes=[$lanes]', d['ms_per_step'])" 2>&1 | tail -1 | tee -a $O/ab3.txt
done; done
for rep in 1 2; do for lanes in "1,3,4" "0,1,3,4,5"; do
  HUGS_STEP_GRAPH_LANES="$lanes" timeout 300 python bench.py --rays-per-gpu 128 --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('128 lanes=[$lanes]', d['ms_per_step'])" 2>&1 | tail -1 | tee -a $O/ab3.txt
  HUGS_STEP_GRAPH_LANES="$lanes" timeout 300 python bench.py --config ref360 --steps 10 --warmup 3 --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('ref360 lanes=[$lanes]', d['ms_per_step'])" 2>&1 | tail -1 | tee -a $O/ab3.txt
done; done
