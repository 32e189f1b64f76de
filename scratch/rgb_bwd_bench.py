import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev='cuda'
for (M,H,dt,td) in ((2097152,256,2,torch.float16),(131072,128,1,torch.bfloat16)):
  Hact=torch.randn(M,H,device=dev).to(td); W=torch.randn(H,3,device=dev)*0.1; rgb=torch.rand(M,3,device=dev); d=torch.randn(M,3,device=dev)
  G=torch.empty(M,H,device=dev,dtype=td); dW=torch.empty(H,3,device=dev); db=torch.empty(3,device=dev)
  ws=torch.empty(L.lib().cdll.hugs_rgb_bwd_ws_bytes()//4,device=dev)
  for _ in range(3): L.call('hugs_rgb_bwd',dt,M,H,Hact,H,W,rgb,d,0.001,G,H,dW,db,ws)
  torch.cuda.synchronize(); t0=time.perf_counter()
  for _ in range(20): L.call('hugs_rgb_bwd',dt,M,H,Hact,H,W,rgb,d,0.001,G,H,dW,db,ws)
  torch.cuda.synchronize(); us=(time.perf_counter()-t0)/20*1e6
  print(f'hugs_rgb_bwd M={M} H={H}: {us:.1f} us = {(2*M*H*2+M*24)/us/1e6:.2f} TB/s')
