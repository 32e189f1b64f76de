import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
M,N,K=131072,1024,1024
A=torch.randn(M,K,device=dev).bfloat16(); Bt=(torch.randn(N,K,device=dev)/32).bfloat16(); bias=torch.randn(N,device=dev)
for v in sys.argv[1:]:
    L._LIB=None; L.LIB_PATH=os.path.join(os.path.dirname(L.LIB_PATH), f'libhugs_v{v}.so')
    for mode in (0, 3):
        out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
        f=lambda: L.call('hugs_gemm_nt_tiles', mode, 1,M,N,K,0,A,K,None,0,Bt,K,bias,None,1,0,1,None,0,None,None,out,N)
        res=[]
        for rep in range(3):
            for _ in range(10): f()
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.append(2*M*N*K/(e0.elapsed_time(e1)/20*1e-3)/1e12)
        print('stagger',v,'mode',mode,[f'{r:.0f}' for r in res])
