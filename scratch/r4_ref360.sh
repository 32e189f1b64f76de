mkdir -p gpurun_out/r4d
python bench.py --config ref360 --min-time 4 --steps 10 --warmup 3 > gpurun_out/r4d/ref360.json 2> gpurun_out/r4d/ref360.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d/ref360.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ['value','ms_per_step','host_enqueue_ms_per_step','step_mfma_frac','step_graph']})
for k,v in d['instep_gemm_shapes_count_avg_us'].items(): print('   ',k,v)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /root/repo/gpurun_out/r4d/trace -o t -- python /root/repo/bench.py --config ref360 --steps 4 --warmup 3 --min-time 0 --no-cpu-baseline > /root/repo/gpurun_out/r4d/prof.log 2>&1
cd /root/repo
STEP=3 python scratch/timeline.py gpurun_out/r4d/trace seq > gpurun_out/r4d/step_timeline_ref360.txt 2>&1
rm -rf gpurun_out/r4d/trace
head -24 gpurun_out/r4d/step_timeline_ref360.txt
