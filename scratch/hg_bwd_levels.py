"""Per-level cost of the field grid's table gradient (k_hashgrid_bwd + the LDS levels) at cfg5: the arguments of the step's own
hugs_hashgrid_bwd calls are captured, then replayed with the output gradient of ONE level at a time (the kernel skips a level whose
gradient columns are zero in a whole wave)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_hugs_amd import _lib as L
from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
from nerf_hugs_amd.nerfacto.configs import PHOTOTOURISM_NERFACTO_BASE as CFG5
dev = 'cuda'
model = NerfactoModel(NerfactoConfig(**CFG5), device=dev, compute_dtype=os.environ.get('DT', 'fp16'), seed=20200823)
N = 16384
g = torch.Generator(device=dev).manual_seed(100)
d = torch.randn(N, 3, generator=g, device=dev); d = d / d.norm(dim=-1, keepdim=True)
batch = dict(origin=(torch.rand(N, 3, generator=g, device=dev) - 0.5) * 0.6, direction=d, viewdir=d, near=torch.full((N,), 0.05, device=dev),
             far=torch.full((N,), 3.0, device=dev), embed_idx=torch.randint(0, 3500, (N,), generator=g, device=dev).int(),
             bg_rgb=torch.ones(N, 3, device=dev), rgb=torch.rand(N, 3, generator=g, device=dev))
draws = lambda: [torch.rand(N, generator=g, device=dev) for _ in range(3)]
for _ in range(3): model.train_step(batch, u01=draws())
caps = []
orig = L.call
def spy(name, *a):
  if name == 'hugs_hashgrid_bwd':
    caps.append(tuple(x.clone() if torch.is_tensor(x) and x.dtype != torch.float32 or False else x for x in a))
  return orig(name, *a)
L.call = spy
model.train_step(batch, u01=draws()); torch.cuda.synchronize()
L.call = orig
def timeit(fn, n=10):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for a in caps:
  n, nl, F = a[0], a[1], a[2]
  x01, dX0, dtc, pitch, table = a[6], a[7], a[8], a[9], a[10]
  res = np.asarray(a[4].cpu() if torch.is_tensor(a[4]) else a[4]).reshape(-1)[:nl] if not isinstance(a[4], int) else None
  scratch = torch.zeros_like(table)
  full = timeit(lambda: orig('hugs_hashgrid_bwd', *a[:10], scratch))
  nzfrac = float((dX0.float().abs().sum(-1) != 0).float().mean())
  print(f'grid: {n} samples x {nl} levels x {F} features, dtype code {dtc}: all levels {full:.1f} us; rows with any gradient {nzfrac:.3f}')
  tot = 0.0
  for l in range(nl):
    z = torch.zeros_like(dX0); z[:, l * F:(l + 1) * F] = dX0[:, l * F:(l + 1) * F]
    t = timeit(lambda: orig('hugs_hashgrid_bwd', *a[:7], z, *a[8:10], scratch))
    nzl = float((z[:, l * F:(l + 1) * F].float().abs().sum(-1) != 0).float().mean())
    tot += t
    print(f'  level {l:2d}: {t:7.1f} us   nonzero rows {nzl:.3f}')
  print(f'  sum of single-level runs {tot:.1f} us')
