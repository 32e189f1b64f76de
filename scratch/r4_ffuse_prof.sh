ROOT=$PWD
for v in "1 1" "1 0"; do set -- $v
cd /tmp && export TMPDIR=/tmp
HUGS_NF_FIELD_FUSE=$1 HUGS_NF_FIELD_FUSE_BWD=$2 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/ffprof -o t -- python $ROOT/bench.py --config cfg5 --min-time 0 --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
cd $ROOT
python - $1 $2 <<'PY'
import csv, glob, collections, sys
f = glob.glob(f'gpurun_out/ffprof/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in agg.values())
print('FUSE', sys.argv[1], 'BWD', sys.argv[2])
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k[:90]:90s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
print(f"TOTAL GPU kernel time {tot/1e6:.3f} ms over {sum(len(v) for v in agg.values())} dispatches")
PY
rm -rf gpurun_out/ffprof
done
