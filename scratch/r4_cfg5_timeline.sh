ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/c5tl -o t -- python $ROOT/bench.py --config cfg5 --min-time 0 --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $ROOT
STEP_MARK=k_nf_adam_amp STEP=14 python scratch/timeline.py gpurun_out/c5tl seq > gpurun_out/r4_cfg5_timeline.txt 2>&1
rm -rf gpurun_out/c5tl
head -120 gpurun_out/r4_cfg5_timeline.txt
