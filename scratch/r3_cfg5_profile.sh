#!/bin/bash
# round 3: rocprofv3 kernel trace of the cfg5 (nerfacto, fp16 mode) bench command -> per-kernel table
cd "$(dirname "$0")/.."
ROOT=$PWD
mkdir -p gpurun_out/r3c
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r3c/trace -o t -- python $ROOT/bench.py --config cfg5 --steps 8 --warmup 3 --min-time 0 > $ROOT/gpurun_out/r3c/bench_under_rocprof.log 2>&1
cd $ROOT
python - <<'PY' > gpurun_out/r3c/kernel_stats.txt
import csv, glob, collections
f = glob.glob('gpurun_out/r3c/trace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print("# rocprofv3 --kernel-trace of: python bench.py --config cfg5 --steps 8 --warmup 3 --min-time 0   (nerfacto, fp16 mode; 11 train steps + 3 event-bracketed roofline steps)")
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
print(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
PY
rm -rf gpurun_out/r3c/trace
python bench.py --config cfg5 > gpurun_out/r3c/cfg5.json 2>/dev/null
