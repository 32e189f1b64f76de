#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5z; O=gpurun_out/r5z; ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/trace128 -o t -- python $ROOT/bench.py --steps 10 --warmup 5 --min-time 0 --no-cpu-baseline --rays-per-gpu 128 > $ROOT/$O/rocprof128.log 2>&1
cd $ROOT
STEP=9 python scratch/timeline.py $O/trace128 seq > $O/step_timeline_128.txt 2>&1
rm -rf $O/trace128
