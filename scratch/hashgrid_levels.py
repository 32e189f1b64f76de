import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd.nerfacto import encodings as E
n = 16384 * 48
x = torch.rand(n, 3, device='cuda')
for res in (16, 23, 31, 43, 81, 154, 512, 2048):
  g = E.HashGrid(n_levels=1, base_resolution=res, max_resolution=res, log2_hashmap_size=19)
  d_out = torch.randn(n, 2, device='cuda').bfloat16(); d_table = torch.zeros_like(g.table)
  out = torch.empty(n, 2, device='cuda', dtype=torch.bfloat16)
  res_t = {}
  for name, fn in (('fwd', lambda: g.forward(x, out=out)), ('bwd', lambda: g.backward(x, d_out, d_table))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    res_t[name] = e0.elapsed_time(e1) / 10 * 1e3
  print(f'res {res:5d} entries {g.n_entries:7d}: fwd {res_t["fwd"]:.0f} us  bwd {res_t["bwd"]:.0f} us')
