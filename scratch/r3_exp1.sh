#!/bin/bash
# round-3 experiment 1: per-XCD stagger and plain-vs-nontemporal output stores on the persistent NT kernel (trace builds)
cd "$(dirname "$0")/.."
for lib in trace trace_plain; do
  for cfg in "0 0" "-1 1" "-1 2" "-1 4" "-1 7" "8 4"; do
    set -- $cfg
    echo "== lib=$lib SG=$1 SI=$2"
    HUGS_LIB_PATH=$PWD/scratch/libhugs_$lib.so SG=$1 SI=$2 python scratch/ntp_trace.py 2>&1 | tail -7
  done
done
