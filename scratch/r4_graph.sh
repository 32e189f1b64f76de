#!/bin/bash
mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_step_graph.py -q -x 2>&1 | tail -15
for r in 128 256 512 1024; do
for g in 0 1; do
  python bench.py --rays-per-gpu $r --min-time 2 --no-cpu-baseline --step-graph $g > gpurun_out/r4c/sb_${r}_g$g.json 2> gpurun_out/r4c/sb_${r}_g$g.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/r4c/sb_${r}_g$g.json').read().strip().splitlines()[-1])
  print('rays $r graph $g:', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], 'rays/s', d['value'], 'graph_active', d['step_graph'])
except Exception as e:
  print('rays $r graph $g FAILED', e); print(open('gpurun_out/r4c/sb_${r}_g$g.err').read()[-1500:])
PY
done
done
