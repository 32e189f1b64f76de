import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd.nerfacto import encodings as E
g = E.HashGrid()          # nerfacto main field: 16 levels x 2, 2^19, 16 -> 2048
for n in (16384 * 48, 16384 * 256):
  x = torch.rand(n, 3, device='cuda')
  out = torch.empty(n, 32, device='cuda', dtype=torch.bfloat16)
  d_out = torch.randn(n, 32, device='cuda').bfloat16(); d_table = torch.zeros_like(g.table)
  for name, fn in (('fwd', lambda: g.forward(x, out=out)), ('bwd', lambda: g.backward(x, d_out, d_table))):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f'hashgrid {name} n={n}: {t*1e3:.0f} us = {n/t/1e6:.2f} G samples/s, {n*16*8*8/t/1e9:.2f} TB/s of 8-byte gathers')
