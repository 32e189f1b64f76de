import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
def t(f, n=10):
  for _ in range(3): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for M, in_dim in ((8388608, 10), (4194304, 14)):
  H = 64
  X = torch.randn(M, 16, device=dev).half(); W0 = torch.randn(128, 128, device=dev) * 0.3; b0 = torch.randn(128, device=dev) * 0.1
  w1 = torch.randn(128, 128, device=dev) * 0.3; b1 = torch.zeros(1, device=dev); sel = torch.ones(M, device=dev)
  raw, dens = torch.empty(M, device=dev), torch.empty(M, device=dev)
  dd = torch.randn(M, device=dev) * 1e-3; dX = torch.empty(M, 16, device=dev, dtype=torch.float16)
  g = [torch.empty(128, 128, device=dev), torch.empty(128, device=dev), torch.empty(128, 128, device=dev), torch.empty(1, device=dev)]
  ws = torch.empty(L.lib().cdll.hugs_nf_prop_ws_bytes(in_dim) // 4, device=dev)
  print(M, in_dim, 'fwd', f"{t(lambda: L.call('hugs_nf_prop_fwd', M, in_dim, H, 2, X, 16, W0, 128, b0, w1, 128, b1, sel, raw, dens, 0, -1.0)):.1f} us",
        'bwd', f"{t(lambda: L.call('hugs_nf_prop_bwd', M, in_dim, H, 2, X, 16, W0, 128, b0, w1, 128, raw, sel, dd, dX, g[0], g[1], g[2], g[3], ws, 0, 0, -1.0)):.1f} us", flush=True)
