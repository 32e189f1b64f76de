#!/bin/bash
mkdir -p gpurun_out/r4c
for r in 128 1024; do
for lanes in "" "1" "1,3" "0,1,3"; do
  HUGS_STEP_GRAPH_LANES=$lanes timeout 300 python -X faulthandler bench.py --rays-per-gpu $r --min-time 2 --no-cpu-baseline --step-graph 1 > gpurun_out/r4c/gl.json 2> gpurun_out/r4c/gl.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/r4c/gl.json').read().strip().splitlines()[-1])
  print('rays $r graph lanes [$lanes]:', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], 'graph_active', d['step_graph'])
except Exception as e:
  print('rays $r lanes [$lanes] FAILED', e); print(open('gpurun_out/r4c/gl.err').read()[-600:])
PY
done
  python bench.py --rays-per-gpu $r --min-time 2 --no-cpu-baseline --step-graph 0 > gpurun_out/r4c/gl.json 2> gpurun_out/r4c/gl.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4c/gl.json').read().strip().splitlines()[-1])
print('rays $r eager:', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'])
PY
done
