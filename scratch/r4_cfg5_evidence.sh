#!/bin/bash
# round-4 cfg5 evidence after the fused field kernels: bench lines (fp16 / bf16, per-kernel table), layer-by-layer A/B, rocprofv3 kernel stats
cd "$(dirname "$0")/.."
ROOT=$PWD
O=gpurun_out/r4c5; mkdir -p $O
python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 > $O/cfg5.json 2>/dev/null
python bench.py --config cfg5 --dtype bf16 --min-time 3 --steps 10 --warmup 5 > $O/cfg5_bf16.json 2>/dev/null
for v in "0 0" "1 0" "1 1"; do set -- $v
  HUGS_NF_FIELD_FUSE=$1 HUGS_NF_FIELD_FUSE_BWD=$2 python bench.py --config cfg5 --min-time 3 --steps 10 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 fp16 HUGS_NF_FIELD_FUSE=$1 HUGS_NF_FIELD_FUSE_BWD=$2: ms_per_step', d['ms_per_step'], 'rays/s', d['value'])" >> $O/cfg5_fuse_ab.txt
done
python scratch/ffuse_bench.py 2>/dev/null | grep -v amdgpu >> $O/cfg5_fuse_ab.txt
python scratch/fbwd_bench.py 2>/dev/null | grep -v amdgpu >> $O/cfg5_fuse_ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/trace -o t -- python $ROOT/bench.py --config cfg5 --min-time 0 --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $ROOT
python - <<'PY' > gpurun_out/r4c5/cfg5_kernel_stats.txt
import csv, glob, collections
f = glob.glob('gpurun_out/r4c5/trace/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print("rocprofv3 --kernel-trace --stats -- python bench.py --config cfg5 --min-time 0 --steps 10 --warmup 5 (fp16, fused field kernels on)")
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:90]:90s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
print(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
PY
rm -rf $O/trace
cat $O/cfg5_fuse_ab.txt
python - <<'PY'
import json
for f in ('cfg5','cfg5_bf16'):
  d=json.loads(open(f'gpurun_out/r4c5/{f}.json').read().strip().splitlines()[-1])
  print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])
  for e in d['instep_kernels'][:8]: print('   ', e['ms_per_step'], e['bound'], e['frac'], e['kernel'][:80])
PY
