#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5r; O=gpurun_out/r5r
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_all.txt
python bench.py --no-cpu-baseline --min-time 4 2>/dev/null | tail -1 > $O/bench.json
python -c "import json;d=json.load(open('$O/bench.json'));print('cfg2', d['ms_per_step'], d['value'], d['step_mfma_frac'], d['roofline']['frac'])"
