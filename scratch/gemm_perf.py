import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
def check(M,N,K1,K2,relu=1,mask=False,r1=False,rb=False):
    A1=torch.randn(M,K1,device=dev).bfloat16(); A2=torch.randn(M,K2,device=dev).bfloat16() if K2 else None
    Bt=(torch.randn(N,K1+K2,device=dev)/(K1+K2)**0.5).bfloat16(); bias=torch.randn(N,device=dev)
    rbt=torch.randn(M//64,N,device=dev) if rb else None; mk=torch.randn(M,N,device=dev).bfloat16() if mask else None
    rr=torch.randn(M,device=dev) if r1 else None; rc=torch.randn(N,device=dev) if r1 else None
    outs=[]
    for small in (0,1,2):
        out=torch.empty(M,N,device=dev,dtype=torch.bfloat16)
        L.call('hugs_gemm_nt_tiles', small, 1,M,N,K1,K2,A1,K1,A2,K2,Bt,K1+K2,bias,rbt,64,N,relu,mk,N,rr,rc,out,N)
        outs.append(out)
    A=torch.cat([A1,A2],1) if K2 else A1
    ref=A.double()@Bt.double().T+bias.double()
    if rb: ref+=rbt.double().repeat_interleave(64,0)
    if r1: ref+=rr.double()[:,None]*rc.double()[None]
    if relu: ref=ref.clamp(min=0)
    if mask: ref=ref*(mk.double()>0)
    print(f'M={M} N={N} K={K1}+{K2} relu={relu} mask={mask} r1={r1} rb={rb}: err256 {(outs[0].double()-ref).abs().max().item():.3e} err128 {(outs[1].double()-ref).abs().max().item():.3e} equal {torch.equal(outs[0],outs[1])} {torch.equal(outs[0],outs[2])}')
check(256,256,128,0); check(1024,1024,1024,512); check(512,256,512,0,relu=0,mask=True,r1=True); check(768,256,256,0,rb=True); check(2048,1024,256,0); check(512,128,1024,512)
def perf(M,N,K,small,mask=False):
    A=torch.randn(M,K,device=dev).bfloat16(); Bt=(torch.randn(N,K,device=dev)/32).bfloat16(); bias=torch.zeros(N,device=dev)
    out=torch.empty(M,N,device=dev,dtype=torch.bfloat16); mk=torch.randn(M,N,device=dev).bfloat16() if mask else None
    f=lambda: L.call('hugs_gemm_nt_tiles', small, 1,M,N,K,0,A,K,None,0,Bt,K,bias,None,1,0,0 if mask else 1,mk,N,None,None,out,N)
    for _ in range(3): f()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    dt=e0.elapsed_time(e1)/20*1e-3
    print(f'perf M={M} N={N} K={K} small={small} mask={mask}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF')
for small in (1,2,0):
    perf(131072,1024,1024,small); perf(131072,1024,1024,small,mask=True); perf(131072,1024,512,small); perf(65536,256,256,small); perf(131072,256,1024,small); perf(131072,1024,256,small)
