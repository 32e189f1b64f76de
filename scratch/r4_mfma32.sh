# v_mfma_f32_32x32x16_bf16 timing stand-in in the shipped persistent NT kernel (scratch/libhugs_mfma32.so =
# hugs_gemm.hip built with -DHUGS_MFMA32_STANDIN) against the shipped 16x16x32 kernel, same box, GEMM alone and in the cfg2 step
cat > /tmp/perf_only.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
def perf(M,N,K):
    A=torch.randn(M,K,device=dev).bfloat16(); Bt=(torch.randn(N,K,device=dev)/32).bfloat16(); bias=torch.zeros(N,device=dev)
    out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
    f=lambda: L.call('hugs_gemm_nt', 1,M,N,K,0,A,K,None,0,Bt,K,bias,None,1,0,1,None,0,None,None,out,N)
    for _ in range(5): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    dt=e0.elapsed_time(e1)/50*1e-3
    print(f'  nt relu M={M} N={N} K={K}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF')
for rep in range(2):
    perf(131072,1024,1024); perf(131072,1024,256); perf(262144,1024,1024)
PY
for rep in 1 2; do for l in nerf-hugs_amd/csrc/libhugs_hip.so scratch/libhugs_mfma32.so; do echo "== $l"; HUGS_LIB_PATH=$PWD/$l python /tmp/perf_only.py 2>&1 | grep -v amdgpu; done; done
for rep in 1 2; do for l in nerf-hugs_amd/csrc/libhugs_hip.so scratch/libhugs_mfma32.so; do
  HUGS_LIB_PATH=$PWD/$l python bench.py --min-time 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
try:
  d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 step $l', d['ms_per_step'], d['value'], 'fwd NT frac', d.get('roofline',{}).get('frac'), 'avg_us', d.get('roofline',{}).get('avg_us'))
except Exception as e: print('cfg2 step $l failed', e)"
done; done
