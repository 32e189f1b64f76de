#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5r
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r5r/trace -o t -- python $R/bench.py --config ref360 --steps 4 --warmup 3 --min-time 0 --no-cpu-baseline --batch-pool 4 > $R/gpurun_out/r5r/prof.log 2>&1
cd $R
STEP=3 python scratch/timeline.py gpurun_out/r5r/trace seq > gpurun_out/r5r/step_timeline_ref360.txt 2>&1
rm -rf gpurun_out/r5r/trace
head -30 gpurun_out/r5r/step_timeline_ref360.txt
