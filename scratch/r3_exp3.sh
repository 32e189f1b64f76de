#!/bin/bash
# round-3 experiment 3: iterations 1-3 of every tile in the quartered form (default build) vs the round-2 form (libhugs_unq);
# both with the four-slot kernel (HUGS_NT_PERS5=0) and the quartered build also with the five-slot kernel
cd "$(dirname "$0")/.."
for cfg in "trace 0" "trace_unq 0" "trace 1"; do
  set -- $cfg
  echo "== lib=$1 HUGS_NT_PERS5=$2"
  HUGS_NT_PERS5=$2 HUGS_LIB_PATH=$PWD/scratch/libhugs_$1.so python scratch/ntp_trace.py 2>&1 | tail -7 | head -5
done
for rep in 1 2; do for cfg in "csrc/libhugs_hip 0" "../scratch/libhugs_unq 0" "csrc/libhugs_hip 1"; do
  set -- $cfg
  HUGS_NT_PERS5=$2 HUGS_LIB_PATH=$PWD/nerf-hugs_amd/$1.so python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 pers5=$2', d['value'], d['ms_per_step'], d['value_min'], d['value_max'], d['roofline']['avg_us'], d['roofline']['frac'], [k['avg_us'] for k in d['instep_kernels']])"
done; done
