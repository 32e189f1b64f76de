"""Where does the data-dependent power of the trunk GEMM go: the matrix pipes or the data movement?  Same launch, operand
DATA varied per operand: A (activations, 99 % of the bytes moved) and W (weights)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, N, K = 131072, 1024, 1024
def fill(kind, shape, scale=1.0):
  if kind == 'zeros': return torch.zeros(shape, device=dev).bfloat16()
  if kind == 'ones': return torch.full(shape, 1.0, device=dev).bfloat16()
  if kind == 'relu': return (torch.randn(shape, device=dev) * scale).clamp_min(0).bfloat16()       # post-relu activations: half zeros
  if kind == 'rowconst': return (torch.randn(shape[0], 1, device=dev) * scale).expand(shape).contiguous().bfloat16()   # every element of a row equal
  return (torch.randn(shape, device=dev) * scale).bfloat16()
def perf(ka, kw, reps=150):
  A = fill(ka, (M, K)); Bt = fill(kw, (N, K), 1 / 32); bias = torch.zeros(N, device=dev)
  out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  f = lambda: L.call('hugs_gemm_nt_tiles', 0, 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, N, None, None, out, N)
  for _ in range(30): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize()
  dt = e0.elapsed_time(e1) / reps * 1e-3
  print(f'A {ka:9s} W {kw:9s}: {dt*1e6:7.1f} us {2*M*N*K/dt/1e12:6.0f} TF', flush=True)
for ka, kw in (('zeros', 'zeros'), ('randn', 'zeros'), ('zeros', 'randn'), ('randn', 'randn'), ('relu', 'randn'), ('ones', 'randn'), ('randn', 'ones'),
               ('rowconst', 'randn'), ('randn', 'randn')):
  perf(ka, kw)
