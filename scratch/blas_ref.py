"""Reference point only (NOT used by the product): what torch.matmul (hipBLASLt/rocBLAS) reaches on the trunk
shapes, to know how much headroom the hand-written kernels leave."""
import torch, time
def bench(M,N,K,tr):
  A=torch.randn(M,K,device='cuda').bfloat16(); 
  B=(torch.randn(N,K,device='cuda')/32).bfloat16()
  X=torch.randn(M,N,device='cuda').bfloat16()
  if tr=='nt': f=lambda: A@B.t()
  else: f=lambda: A.t()@X          # [K,M]x[M,N] -> dW
  for _ in range(10): f()
  torch.cuda.synchronize()
  e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20): f()
  e1.record(); torch.cuda.synchronize()
  dt=e0.elapsed_time(e1)/20*1e-3
  print(tr,M,N,K,f'{dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF')
bench(131072,1024,1024,'nt'); bench(131072,1024,1024,'tn'); bench(65536,256,256,'nt'); bench(131072,1024,512,'nt'); bench(131072,1024,1536,'nt')
bench(524288,1024,1024,'nt')
