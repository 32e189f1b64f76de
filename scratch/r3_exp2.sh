#!/bin/bash
# round-3 experiment 2: five-slot persistent NT kernel (HUGS_NT_PERS5=1, default) vs the four-slot one (=0)
cd "$(dirname "$0")/.."
python scratch/pers_perf.py 2>&1 | grep -E "persistent ==|perf" | head -30
for v in 1 0; do
  echo "== trace HUGS_NT_PERS5=$v"
  HUGS_NT_PERS5=$v HUGS_LIB_PATH=$PWD/scratch/libhugs_trace.so python scratch/ntp_trace.py 2>&1 | tail -7
done
for rep in 1 2; do for v in 1 0; do
  HUGS_NT_PERS5=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pers5=$v', d['value'], d['ms_per_step'], d['value_min'], d['value_max'], d['roofline']['avg_us'], d['roofline']['frac'])"
done; done
