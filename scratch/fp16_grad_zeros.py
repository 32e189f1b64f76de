"""cfg5 at the yml sizes: how many entries of the hash-grid input gradients (dX0 of each level) are exactly zero in bf16 and in
fp16 mode (loss scale 65536)?  Zero runs issue no table atomics (csrc/hugs_hashgrid.hip), which is where the fp16 mode's
faster table gradient comes from."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
from nerf_hugs_amd.nerfacto.configs import PHOTOTOURISM_NERFACTO_BASE as YML
dev = 'cuda'; N = 16384
for cdt in ('bf16', 'fp16'):
  model = NerfactoModel(NerfactoConfig(**dict(YML, warmup_steps=10)), compute_dtype=cdt, seed=3)
  g = torch.Generator(device=dev).manual_seed(100)
  d = torch.randn(N, 3, generator=g, device=dev); d = d / d.norm(dim=-1, keepdim=True)
  o = (torch.rand(N, 3, generator=g, device=dev) - 0.5) * 0.6
  batch = dict(origin=o, direction=d, viewdir=d, near=torch.full((N,), 0.05, device=dev), far=torch.full((N,), 3.0, device=dev),
               embed_idx=torch.randint(0, 3500, (N,), generator=g, device=dev).int(), bg_rgb=torch.ones(N, 3, device=dev),
               rgb=(0.5 + 0.5 * torch.sin(3.0 * d + 2.0 * o)).contiguous())
  for i in range(40):
    res = model.train_step(batch, u01=[torch.rand(N, generator=g, device=dev) for _ in range(3)])
    if i in (0, 39):
      out = []
      for name in ('prop0', 'prop1', 'field'):
        key = [k for k in model.ws.bufs if isinstance(k, tuple) and isinstance(k[0], str) and k[0].startswith('dX0') and k[0].endswith(name)]
        t = model.ws.bufs[key[0]]
        nl = model.grids[name].n_output_dims
        z = float((t[:, :nl] == 0).float().mean())
        a = t[:, :nl].float().abs()
        out.append(f'{name}: zero {z:.3f}, median |g| {float(a.flatten()[::97].median()):.2e}, max {float(a.max()):.2e}')
      print(cdt, 'step', i, 'loss', f"{float(res['stats'][1]):.5f}", 'scale', model.loss_scale(), ' | '.join(out), flush=True)
