#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r3c/trace2 -o t -- python $ROOT/bench.py --config cfg5 --steps 6 --warmup 3 --min-time 0 > /dev/null 2>&1
cd $ROOT
STEP=4 STEP_MARK=k_amp_update python scratch/timeline.py gpurun_out/r3c/trace2 seq > gpurun_out/r3c/cfg5_timeline.txt 2>&1
rm -rf gpurun_out/r3c/trace2
head -12 gpurun_out/r3c/cfg5_timeline.txt
