#!/bin/bash
# round 6 evidence set on ONE box: PMC passes of the trunk GEMMs, the bench line, rocprofv3 kernel stats + step timeline of the bench
# command, the other configs, the small-batch table.  Usage (on the GPU box): bash scratch/r6_evidence.sh <tag>
tag=${1:-r06}
cd "$(dirname "$0")/.."
ROOT=$PWD
O=gpurun_out/$tag
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
bash scratch/pmc_run2.sh $tag > $O/pmc.log 2>&1
python scratch/pmc_to_traffic.py $tag > $O/pmc_traffic.log 2>&1 && cp profiles/${tag}_gemm_traffic.json profiles/${tag}_gemm_pmc_raw.json $O/
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/trace -o t -- python $ROOT/bench.py --steps 10 --warmup 5 --min-time 0 --no-cpu-baseline > $ROOT/$O/bench_under_rocprof.log 2>&1
cd $ROOT
python - $O <<'PY' > $O/kernel_stats.txt
import csv, glob, collections, sys
f = glob.glob(f'{sys.argv[1]}/trace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
print(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
PY
STEP=9 python scratch/timeline.py $O/trace seq > $O/step_timeline.txt 2>&1
rm -rf $O/trace
python bench.py > $O/bench_b.json 2>> $O/bench.err      # (again, now with this box's traffic file in place)
for c in cfg3 cfg4 ref360 cfg5; do python bench.py --config $c > $O/${c}_bench.json 2>> $O/bench.err; done
python bench.py --config cfg5 --dtype bf16 > $O/cfg5_bench_bf16.json 2>> $O/bench.err
python - $O <<'PY'
import json, subprocess, sys
out = {}
for r in (128, 256, 512, 1024):
  for g in ('0', '1'):
    p = subprocess.run([sys.executable, 'bench.py', '--rays-per-gpu', str(r), '--step-graph', g, '--no-cpu-baseline', '--min-time', '3'], capture_output=True, text=True)
    try:
      d = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
      out[f'{r}_graph{g}'] = {k: d[k] for k in ('ms_per_step', 'value', 'host_enqueue_ms_per_step', 'step_graph')}
    except Exception as e:
      out[f'{r}_graph{g}'] = {'error': repr(e), 'stderr': p.stderr[-500:]}
json.dump(out, open(f'{sys.argv[1]}/small_batch.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
