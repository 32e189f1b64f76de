#!/bin/bash
# round-4 first GPU call: test suite, headline bench, small-batch table, reference-default shape
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/r4a/pytest.log
python bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; tail -c 600 gpurun_out/r4a/bench.json
for r in 128 256 512 1024; do
  python bench.py --rays-per-gpu $r --min-time 3 --no-cpu-baseline > gpurun_out/r4a/small_$r.json 2> gpurun_out/r4a/small_$r.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4a/small_$r.json').read().strip().splitlines()[-1])
print($r, d['ms_per_step'], d['host_enqueue_ms_per_step'], d['value'])
PY
done
python bench.py --config ref360 --min-time 4 --steps 10 --warmup 3 > gpurun_out/r4a/ref360.json 2> gpurun_out/r4a/ref360.err; tail -c 1500 gpurun_out/r4a/ref360.json; tail -5 gpurun_out/r4a/ref360.err
