#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5p2; O=gpurun_out/r5p2; ROOT=$PWD
for v in nerfw; do for g in 1 0; do
  cd /tmp && export TMPDIR=/tmp
  HUGS_STEP_GRAPH=$g timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/trace_$v -o t -- python $ROOT/scratch/variant_prof.py $v > $ROOT/$O/rocprof_$v.log 2>&1
  cd $ROOT
  STEP=9 python scratch/timeline.py $O/trace_$v seq > $O/timeline_${v}_g$g.txt 2>&1
  rm -rf $O/trace_$v
  head -3 $O/timeline_${v}_g$g.txt
done; done
