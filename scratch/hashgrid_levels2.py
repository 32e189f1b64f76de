"""scratch: per-level cost of the hash-grid table-gradient scatter on ray-ordered samples (cfg5's first proposal level:
16384 rays x 512 samples, 5 levels 16..128, 2^17 entries; and the field grid's levels at 128 samples per ray)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd.nerfacto import encodings as E
torch.manual_seed(0)
def positions(N, S):
  o = torch.rand(N, 3, device='cuda') * 0.2 + 0.4
  d = torch.randn(N, 3, device='cuda'); d = d / d.norm(dim=-1, keepdim=True)
  t = torch.linspace(0, 0.45, S, device='cuda')[None, :, None]
  return (o[:, None] + d[:, None] * t).clamp(0, 1).reshape(-1, 3).contiguous()
def run(x, res, log2):
  g = E.HashGrid(n_levels=1, base_resolution=res, max_resolution=res, log2_hashmap_size=log2)
  n = x.shape[0]
  d_out = torch.randn(n, 2, device='cuda').bfloat16(); d_table = torch.zeros_like(g.table)
  fn = lambda: g.backward(x, d_out, d_table)
  for _ in range(2): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / 5 * 1e3, g.n_entries
x = positions(16384, 512)
for res in (16, 27, 45, 76, 128): 
  t, ne = run(x, res, 17); print(f'prop0 S=512 res {res:5d} entries {ne:7d}: bwd {t:8.0f} us')
x = positions(16384, 128)
for res in (16, 32, 64, 128, 256, 512, 1024, 2048):
  t, ne = run(x, res, 19); print(f'field S=128 res {res:5d} entries {ne:7d}: bwd {t:8.0f} us')
