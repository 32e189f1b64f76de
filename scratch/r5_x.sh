#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5x; O=gpurun_out/r5x; rm -f $O/ab.txt
for rep in 1 2; do for a in 0 1; do for b in 0 1; do
  HUGS_INTERLEVEL_ON_PROP=$a HUGS_COMPOSITE_RAW=$b python bench.py --no-cpu-baseline --min-time 2.5 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('il_prop=$a comp_raw=$b', d['ms_per_step'])" | tee -a $O/ab.txt
done; done; done
for a in 0 1; do for b in 0 1; do
  HUGS_INTERLEVEL_ON_PROP=$a HUGS_COMPOSITE_RAW=$b python bench.py --rays-per-gpu 128 --no-cpu-baseline --min-time 2 2>/dev/null | tail -1 | python -c "import sys,json;d=json.loads(sys.stdin.read());print('128: il_prop=$a comp_raw=$b', d['ms_per_step'])" | tee -a $O/ab.txt
done; done
