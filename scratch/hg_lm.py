import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
from nerf_hugs_amd.nerfacto.encodings import HashGrid
dev = 'cuda'
for (nl, lh, maxres, n) in ((16, 19, 2048, 2097152), (5, 17, 128, 8388608), (7, 17, 256, 4194304)):
  g = HashGrid(nl, 2, lh, 16, None, maxres, device=dev)
  gen = torch.Generator(device=dev).manual_seed(0)
  # samples along rays (consecutive samples are neighbours in space), like the model's
  o = torch.rand(n // 128, 1, 3, generator=gen, device=dev) * 0.4 + 0.3
  d = torch.randn(n // 128, 1, 3, generator=gen, device=dev); d = d / d.norm(dim=-1, keepdim=True)
  t = torch.linspace(0, 0.3, 128, device=dev)[None, :, None]
  x = (o + d * t).clamp(0, 1).reshape(-1, 3).contiguous()
  th = ((torch.rand(g.n_entries, 2, generator=gen, device=dev) * 2 - 1) * 1e-1).half()
  o_, r_, s_ = g._tables()
  out = torch.zeros(n, 32, device=dev, dtype=torch.float16)
  f = lambda: L.call('hugs_hashgrid_fwd_t', n, nl, 2, o_, r_, s_, x, th, 2, 2, 32, out)
  for _ in range(3): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10): f()
  e1.record(); torch.cuda.synchronize()
  print(f'levels {nl} log2 {lh} n {n}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  checksum {float(out.float().sum()):.4f}', flush=True)
  # backward: gradient rows in half
  dX = (torch.randn(n, 32, generator=gen, device=dev) * 1e-2).half()
  gt = torch.zeros(g.n_entries, 2, device=dev)
  fb = lambda: L.call('hugs_hashgrid_bwd', n, nl, 2, o_, r_, s_, x, dX, 2, 32, gt)
  for _ in range(2): fb()
  torch.cuda.synchronize()
  gt.zero_(); fb(); torch.cuda.synchronize(); cs = float(gt.double().abs().sum())
  e0.record()
  for _ in range(5): fb()
  e1.record(); torch.cuda.synchronize()
  print(f'   bwd: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us  checksum {cs:.4f}', flush=True)
