#!/bin/bash
# round 5: rocprofv3 kernel trace of the bench command (per-kernel table + one step's timeline), default and 128-ray step
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out/r5p
for tag in 1024 128; do
  extra=""; [ $tag = 128 ] && extra="--rays-per-gpu 128"
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r5p/trace$tag -o t -- python $ROOT/bench.py --steps 10 --warmup 5 --min-time 0 --no-cpu-baseline $extra > $ROOT/gpurun_out/r5p/bench_under_rocprof_$tag.log 2>&1
  cd $ROOT
  python - $tag <<'PY' > gpurun_out/r5p/kernel_stats_$tag.txt
import csv, glob, collections, sys
f = glob.glob(f'gpurun_out/r5p/trace{sys.argv[1]}/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
print(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
PY
  STEP=9 python scratch/timeline.py gpurun_out/r5p/trace$tag seq > gpurun_out/r5p/step_timeline_$tag.txt 2>&1
  rm -rf gpurun_out/r5p/trace$tag
done
