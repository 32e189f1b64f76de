import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
def run(Mr,Kc,N,ns,small,check=True,bias=True):
    X=torch.randn(Mr,Kc,device=dev).bfloat16(); G=torch.randn(Mr,N,device=dev).bfloat16()
    dW=torch.empty(Kc,N,device=dev); db=torch.empty(N,device=dev) if bias else None
    ws=torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(Kc,N,ns)//4,device=dev)
    f=lambda: L.call('hugs_gemm_tn_tiles', small, 1,Mr,Kc,N,ns,X,Kc,G,N,dW,db,ws)
    f(); msg=''
    if check:
        ref=X.double().T@G.double()
        msg=f'dW err {(dW.double()-ref).abs().max().item():.2e} (scale {ref.abs().max().item():.0f})'
        if bias: msg+=f' db err {(db.double()-G.double().sum(0)).abs().max().item():.2e}'
    for _ in range(2): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    dt=e0.elapsed_time(e1)/10*1e-3
    print(f'rows={Mr} Kc={Kc} N={N} split={ns} small={small}: {dt*1e6:.1f} us {2*Mr*Kc*N/dt/1e12:.0f} TF  {msg}')
run(8192,256,256,4,0); run(8192,256,256,4,1); run(16384,512,1024,8,0); run(4096,1024,256,16,0,bias=False)
for small in (1,0):
    run(131072,1024,1024,8 if small else 16,small,check=False); run(131072,512,1024,16 if small else 32,small,check=False); run(65536,256,256,64 if small else 256,small,check=False); run(65536,512,256,64 if small else 128,small,check=False); run(131072,1024,256,32 if small else 64, small, check=False)
