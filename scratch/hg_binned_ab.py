"""A/B of the hash-grid table gradient at cfg5: L2 atomic scatter (HUGS_HG_BINNED=0) vs the segmented reduction by table slot (default), on
the arguments of the step's own three hugs_hashgrid_bwd_ws calls: time per call and agreement of the two gradients; then the whole step."""
import os, sys, time
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
  if name == 'hugs_hashgrid_bwd_ws':
    caps.append(tuple(x.clone() if (torch.is_tensor(x) and k in (6, 7)) else x for k, x in enumerate(a)))
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
  table = a[10]
  out = {}
  for mode in ('0', '1'):
    os.environ['HUGS_HG_BINNED'] = mode
    t = torch.zeros_like(table)
    orig('hugs_hashgrid_bwd_ws', *a[:10], t, *a[11:]); torch.cuda.synchronize()
    scratch = torch.zeros_like(table)
    us = timeit(lambda: orig('hugs_hashgrid_bwd_ws', *a[:10], scratch, *a[11:]))
    out[mode] = (t, us)
  t0, t1 = out['0'][0].double(), out['1'][0].double()
  err = float((t0 - t1).abs().max() / t0.abs().max().clamp(min=1e-30))
  rel = float((t0 - t1).norm() / t0.norm().clamp(min=1e-30))
  print(f'grid {n} samples x {nl} levels: scatter {out["0"][1]:8.1f} us   binned {out["1"][1]:8.1f} us   max err / max {err:.2e}  rel L2 {rel:.2e}  nonzero entries {int((t0 != 0).sum())} vs {int((t1 != 0).sum())}')
for mode in ('0', '1', '0', '1'):
  os.environ['HUGS_HG_BINNED'] = mode
  for _ in range(5): model.train_step(batch, u01=draws())
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(30): model.train_step(batch, u01=draws())
  torch.cuda.synchronize()
  print(f'step, HUGS_HG_BINNED={mode}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms')
