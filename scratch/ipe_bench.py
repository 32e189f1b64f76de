"""scratch: hugs_cast_ipe_fwd timing at the cfg2 shapes (1024 rays x 64 / 128 samples, bf16, 512-wide rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
from nerf_hugs_amd.internal import geopoly
dev = 'cuda'
basis = torch.from_numpy(geopoly.generate_basis('icosahedron', 2).astype(np.float32)).to(dev)
nb = basis.shape[1]
for S in (64, 128):
  N = 1024
  td = torch.sort(torch.rand(N, S + 1, device=dev) * 4 + 0.1, -1).values
  o = torch.randn(N, 3, device=dev) * 0.1; d = torch.randn(N, 3, device=dev); r = torch.full((N, 1), 1e-3, device=dev)
  X = torch.empty(N * S, 512, device=dev, dtype=torch.bfloat16)
  fn = lambda: L.call('hugs_cast_ipe_fwd', N, S, td, o, d, r, basis, nb, 0, 0, 12, 1, 512, X)
  for _ in range(5): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(50): fn()
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / 50 * 1e3
  print(f'S={S}: {us:.1f} us, {N*S*1024/us/1e6:.2f} TB/s written')
