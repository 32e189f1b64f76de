#!/bin/bash
# round-5 evidence run: bench lines (default, small batches eager vs graph, ref360, cfg3/4/5), rocprofv3 kernel trace + step timelines
cd "$(dirname "$0")/.."
ROOT=$PWD
O=gpurun_out/r5e3; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
for r in 128 256 512 1024; do for g in 0 1; do
  python bench.py --rays-per-gpu $r --min-time 3 --no-cpu-baseline --step-graph $g > $O/sb_${r}_g$g.json 2>/dev/null
done; done
python bench.py --config ref360 --min-time 4 --steps 10 --warmup 3 > $O/ref360.json 2>/dev/null
python bench.py --config cfg3 --min-time 3 --steps 10 --warmup 3 > $O/cfg3.json 2>/dev/null
python bench.py --config cfg4 --min-time 3 > $O/cfg4.json 2>/dev/null
python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 > $O/cfg5.json 2>/dev/null
python bench.py --config cfg5 --dtype bf16 --min-time 3 --steps 10 --warmup 5 > $O/cfg5_bf16.json 2>/dev/null
bash scratch/r5_profile.sh > /dev/null 2>&1
cp gpurun_out/r5p/* $O/ 2>/dev/null
python - <<'PY'
import json, glob
O='gpurun_out/r5e3'
rows=[]
for r in (128,256,512,1024):
  row={'rays_per_gpu': r}
  for g in (0,1):
    try:
      d=json.loads(open(f'{O}/sb_{r}_g{g}.json').read().strip().splitlines()[-1])
      row['graph' if g else 'eager']={'ms_per_step': d['ms_per_step'], 'host_enqueue_ms_per_step': d['host_enqueue_ms_per_step'], 'rays_per_s': d['value'], 'step_graph': d['step_graph']}
    except Exception as e:
      row['graph' if g else 'eager']={'error': repr(e)}
  rows.append(row)
json.dump({'what': 'per-rank step of the fixed-global-batch curve on ONE MI355X (bench.py --rays-per-gpu R --step-graph 0|1): wall ms per step, host ms to enqueue a step, rays/s', 'rows': rows}, open(f'{O}/small_batch.json','w'), indent=1)
print(json.dumps(rows, indent=1))
for f in ('bench','ref360','cfg3','cfg4','cfg5','cfg5_bf16'):
  try:
    d=json.loads(open(f'{O}/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d.get('step_mfma_frac'), (d.get('roofline') or {}).get('frac'))
  except Exception as e: print(f, 'FAILED', e)
PY
cp gpurun_out/r5p/* gpurun_out/r5e3/ 2>/dev/null
timeout 600 python scratch/transient_perf.py 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/r5e3/transient_variants.txt
python scratch/ipe_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r5e3/ipe_bench.txt
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r5e3/trace_ref360 -o t -- python $ROOT/bench.py --config ref360 --steps 8 --warmup 4 --min-time 0 --no-cpu-baseline > $ROOT/gpurun_out/r5e3/rocprof_ref360.log 2>&1
cd $ROOT
STEP=9 python scratch/timeline.py gpurun_out/r5e3/trace_ref360 seq > gpurun_out/r5e3/ref360_timeline.txt 2>&1
rm -rf gpurun_out/r5e3/trace_ref360
timeout 900 python scratch/parity_margins.py > gpurun_out/r5e3/parity_margins.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5e3/pytest_full.txt 2>&1; tail -3 gpurun_out/r5e3/pytest_full.txt
