#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5k; O=gpurun_out/r5k
timeout 900 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_data_parallel.py tests/test_gpu_bench_config.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
for r in 128 256; do
  python bench.py --rays-per-gpu $r --min-time 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/sb_$r.json
  python -c "import json;d=json.load(open('$O/sb_$r.json'));print($r, d['ms_per_step'], d['host_enqueue_ms_per_step'], d['step_graph'])"
done
python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 > $O/bench.json
python -c "import json;d=json.load(open('$O/bench.json'));print('cfg2', d['ms_per_step'], d['value'], d['fixed_batch'])"
python bench.py --config cfg4 --no-cpu-baseline --min-time 3 2>/dev/null | tail -1 > $O/cfg4.json
python -c "import json;d=json.load(open('$O/cfg4.json'));print('cfg4', d['ms_per_step'], d['value'])"
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/trace128 -o t -- python $ROOT/bench.py --steps 10 --warmup 5 --min-time 0 --no-cpu-baseline --rays-per-gpu 128 > $ROOT/$O/rocprof128.log 2>&1
cd $ROOT
STEP=9 python scratch/timeline.py $O/trace128 seq > $O/step_timeline_128.txt 2>&1
rm -rf $O/trace128
head -5 $O/step_timeline_128.txt
