"""Driver for the PMC passes (scratch/pmc_run2.sh): the trunk-shape GEMMs of the train step, 4 launches each; round 4: + the
batched weight-gradient launch of the whole NerfMLP trunk (9 items = 8.5 layer-equivalents, 2 reduction pieces per tile)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
M,N,K=131072,1024,1024
A=torch.randn(M,K,device=dev).bfloat16(); Bt=(torch.randn(N,K,device=dev)/32).bfloat16(); bias=torch.zeros(N,device=dev)
out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
G=torch.randn(M,N,device=dev).bfloat16(); dW=torch.empty(K,N,device=dev); db=torch.empty(N,device=dev)
Y=torch.randn(M,N,device=dev).bfloat16()
ws=torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K,N,16)//4,device=dev)
bits=torch.empty(L.lib().cdll.hugs_gemm_nt_bits_bytes(M,N)//4,device=dev,dtype=torch.int32)
for _ in range(4):
    L.call('hugs_gemm_nt',1,M,N,K,0,A,K,None,0,Bt,K,bias,None,1,0,1,None,0,None,None,out,N)      # forward trunk layer, no bit mask (pers<3>)
    L.call('hugs_gemm_nt',1,M,N,K,0,G,K,None,0,Bt,K,None,None,1,0,0,Y,N,None,None,out,N)         # dX masked by the bf16 activation (pers<4>)
    L.call('hugs_gemm_nt_bits',1,M,N,K,0,A,K,None,0,Bt,K,bias,1,None,None,out,N,bits,None)       # forward trunk layer writing 1-bit masks (pers<35>): the step's form
    L.call('hugs_gemm_nt_bits',1,M,N,K,0,G,K,None,0,Bt,K,None,0,None,None,out,N,None,bits)       # dX masked by bits (pers<16>): the step's form
    L.call('hugs_gemm_tn',1,M,K,N,16,A,K,G,N,dW,db,ws)                                             # dW + slab reduce
# round 4: the trunk's batched dW launch as the step issues it (cfg2): 6 x [1024x1024], the skip layer's [1024x1024] + [512x1024], layer 0 [512x1024]
import numpy as np, ctypes
from nerf_hugs_amd.internal import engine as E
Xs = [torch.randn(M, 1024, device=dev).bfloat16() for _ in range(7)]; X0 = torch.randn(M, 512, device=dev).bfloat16()
Gs = [torch.randn(M, 1024, device=dev).bfloat16().clamp_(min=0) for _ in range(8)]
items = [(1024, Xs[i], Gs[i]) for i in range(6)] + [(1024, Xs[6], Gs[6]), (512, X0, Gs[6]), (512, X0, Gs[7])]
dWs = [torch.empty(kc, 1024, device=dev) for kc, _, _ in items]; dbs = [torch.empty(1024, device=dev) for _ in items]
arr = np.zeros(len(items), E._TN_ITEM)
for k, (kc, x, g) in enumerate(items):
    arr[k] = (x.data_ptr(), g.data_ptr(), dWs[k].data_ptr(), dbs[k].data_ptr(), x.shape[1], 1024, M, kc, 1024, 0)
ns = int(L.lib().cdll.hugs_gemm_tn_batch_nsplit(len(items), ctypes.c_void_p(arr.ctypes.data)))
wsb = torch.empty(int(L.lib().cdll.hugs_gemm_tn_batch_ws_bytes(len(items), arr.ctypes.data, ns)) // 4, device=dev)
for _ in range(4):
    L.call('hugs_gemm_tn_batch', 1, len(items), arr.ctypes.data, ns, wsb)
torch.cuda.synchronize()
