"""scratch: K = 128 NT GEMM, 256x256 ring kernel vs the 128x128 kernel (hugs_gemm_nt_tiles tile_mode 1), two M."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
def t(fn, n=30):
  for _ in range(5): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for M in (131072, 2097152):
  for (N, K) in ((256, 128), (128, 256), (256, 256)):
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 16).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    call = lambda force=0: L.call('hugs_gemm_nt_tiles', force, 1, M, N, K, 0, A, K, None, 0, Bt, K, None, None, 1, 0, 0, None, 0, None, None, out, N)
    res = []
    for force in (0, 1):
      res.append(t(lambda: call(force)))
    print(f'M={M} N={N} K={K}: ring {res[0]:.1f} us, 128x128 {res[1]:.1f} us')
