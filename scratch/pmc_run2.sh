#!/bin/bash
# One rocprofv3 pass per counter group (--kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass: 3 + 2 of the
# 4 TCC slots).  Usage: pmc_run2.sh <tag>   ->  gpurun_out/pmc_<tag>.json  (per kernel: mean counter values per launch)
tag=$1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name -o p -- python $GRAFT_REPO_ROOT/scratch/pmc_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, json, shutil
out = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmc_${tag}_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k, c), v in sorted(agg.items()):
            if 'gemm' in k or 'slab' in k:
                out[k][c] = sum(v) / len(v); out[k]['launches'] = len(v)
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        for k, v in agg.items():
            if ('gemm' in k or 'slab' in k) and 'MFMA' in d:
                out[k]['avg_us_profiled'] = sum(v) / len(v) / 1e3
    shutil.rmtree(d, ignore_errors=True)
json.dump(out, open('gpurun_out/pmc_${tag}.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
