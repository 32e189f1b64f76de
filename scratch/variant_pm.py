import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
M,N,K=131072,1024,1024
A=torch.randn(M,K,device=dev).bfloat16(); Bt=(torch.randn(N,K,device=dev)/32).bfloat16(); bias=torch.randn(N,device=dev)
G=torch.randn(M,N,device=dev).bfloat16(); mk=torch.randn(M,N,device=dev).bfloat16()
pm=lambda t: t.view(M, t.shape[1]//256, 256).permute(1,0,2).contiguous()
ref=None; refw=None
for v in sys.argv[1:]:
    L._LIB=None; L.LIB_PATH=os.path.join(os.path.dirname(L.LIB_PATH), f'libhugs_v{v}.so')
    a, g, m = (pm(A), pm(G), pm(mk)) if v=='1' else (A, G, mk)
    out=torch.empty(M,N,device=dev,dtype=torch.bfloat16); out2=torch.empty_like(out)
    f=lambda: L.call('hugs_gemm_nt',1,M,N,K,0,a,K,None,0,Bt,K,bias,None,1,0,1,None,0,None,None,out,N)
    f2=lambda: L.call('hugs_gemm_nt',1,M,N,K,0,a,K,None,0,Bt,K,None,None,1,0,0,m,N,None,None,out2,N)
    dW=torch.empty(K,N,device=dev); db=torch.empty(N,device=dev)
    ws=torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K,N,16)//4,device=dev)
    h=lambda: L.call('hugs_gemm_tn',1,M,K,N,16,a,K,g,N,dW,db,ws)
    res={}
    for name,fn,fl in (('nt',f,2*M*N*K),('nt_mask',f2,2*M*N*K),('tn',h,2*M*N*K)):
        r=[]
        for rep in range(3):
            for _ in range(8): fn()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            r.append(fl/(e0.elapsed_time(e1)/20*1e-3)/1e12)
        res[name]=[f'{x:.0f}' for x in r]
    o = out if v!='1' else out.view(N//256, M, 256).permute(1,0,2).reshape(M,N)
    if ref is None: ref=o.clone(); refw=dW.clone()
    print('pm',v,res,'nt equal',torch.equal(o,ref),'tn equal',torch.equal(dW,refw))
