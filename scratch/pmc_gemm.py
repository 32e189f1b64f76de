"""Driver for the PMC passes (scratch/pmc_run2.sh): the three trunk-shape GEMMs of the train step, 4 launches each."""
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
torch.cuda.synchronize()
