import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
M,N,K=131072,1024,1024
X=torch.randn(M,K,device=dev).bfloat16(); G=torch.randn(M,N,device=dev).bfloat16()
first=None
for v in sys.argv[1:]:
    L._LIB=None; L.LIB_PATH=os.path.join(os.path.dirname(os.path.abspath(L.__file__)), 'csrc', f'libhugs_v{v}.so')
    dW=torch.empty(K,N,device=dev); db=torch.empty(N,device=dev)
    ws=torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K,N,16)//4,device=dev)
    f=lambda: L.call('hugs_gemm_tn',1,M,K,N,16,X,K,G,N,dW,db,ws)
    res=[]
    for rep in range(3):
        for _ in range(5): f()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(15): f()
        e1.record(); torch.cuda.synchronize()
        res.append(2*M*N*K/(e0.elapsed_time(e1)/15*1e-3)/1e12)
    if first is None: first=dW.clone()
    print('variant',v,[f'{r:.0f}' for r in res],'equal' , torch.equal(first,dW))
