#!/bin/bash
# scratch: same-box A/B of library variants on the bench step.  usage: ab.sh v0 v1 v2 ...
for rep in 1 2; do for v in "$@"; do
  HUGS_LIB_PATH=$PWD/scratch/lib$v.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
