#!/bin/bash
# One rocprofv3 pass per counter group (--kernel-trace only), as the microarch guide prescribes.  Usage: pmc_run.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace -f csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name -o p -- python $GRAFT_REPO_ROOT/scratch/pmc_gemm.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$name.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_${tag}_*/')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()):
            if 'gemm' in k[0]:
                print(k[0], k[1], 'mean %.4g over %d' % (sum(v) / len(v), len(v)))
PY
