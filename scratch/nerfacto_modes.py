"""nerfacto (yml sizes, 16384 rays): rgb loss trajectory of the three compute modes on the same learnable synthetic target,
same seed and batches -- fp16 (the reference's enable_amp), bf16, fp32 (parity mode)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
from nerf_hugs_amd.nerfacto.configs import PHOTOTOURISM_NERFACTO_BASE as YML
dev = 'cuda'; N = 16384; STEPS = int(os.environ.get('STEPS', '600'))
print(f'# scratch/nerfacto_modes.py: {STEPS} steps, 16384 rays x (512, 256, 128) samples, phototourism_nerfacto_base.yml sizes, warmup 50 steps; rgb loss (mean of 20 steps)')
res = {}
for cdt in ('fp16', 'bf16', 'fp32'):
  model = NerfactoModel(NerfactoConfig(**dict(YML, warmup_steps=50)), compute_dtype=cdt, seed=3)
  g = torch.Generator(device=dev).manual_seed(100)
  d = torch.randn(N, 3, generator=g, device=dev); d = d / d.norm(dim=-1, keepdim=True)
  o = (torch.rand(N, 3, generator=g, device=dev) - 0.5) * 0.6
  batch = dict(origin=o, direction=d, viewdir=d, near=torch.full((N,), 0.05, device=dev), far=torch.full((N,), 3.0, device=dev),
               embed_idx=torch.randint(0, 3500, (N,), generator=g, device=dev).int(), bg_rgb=torch.ones(N, 3, device=dev),
               rgb=(0.5 + 0.5 * torch.sin(3.0 * d + 2.0 * o)).contiguous())
  ls = []
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for i in range(STEPS):
    r = model.train_step(batch, u01=[torch.rand(N, generator=g, device=dev) for _ in range(3)])
    ls.append(r['stats'][1:2].clone())
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  l = torch.cat(ls).cpu().numpy()
  res[cdt] = l
  print(f'{cdt}: {dt / STEPS * 1e3:.2f} ms/step, loss scale at the end {model.loss_scale():.0f}, finite {bool(np.isfinite(l).all())}')
print('step      fp16        bf16        fp32')
for s in range(0, STEPS, 20):
  print(f'{s:5d}  ' + '  '.join(f'{res[c][s:s + 20].mean():.3e}' for c in ('fp16', 'bf16', 'fp32')))
