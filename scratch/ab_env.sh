#!/bin/bash
# scratch: same-box A/B over env settings: each arg is "NAME=VALUE[,NAME=VALUE...]" or "none"
for rep in 1 2; do for v in "$@"; do
  envs=""; [ "$v" != none ] && envs=$(echo $v | tr ',' ' ')
  env $envs python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_us'], [k['avg_us'] for k in d['instep_kernels']])"
done; done
