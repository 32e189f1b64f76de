"""nerfacto (yml sizes) trained on the analytic sphere scene (tests/analytic_scene.py, scaled by 1/4 into the unit box) with FRESH
16384-ray batches every step, in the three compute modes; validation PSNR on a fixed 4096-ray set.  ADVICE r3: the fp16 mode's
speed comes partly from grid-input gradients that underflow in half -- does the quality follow?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.analytic_scene import scene_batch
from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
from nerf_hugs_amd.nerfacto.configs import PHOTOTOURISM_NERFACTO_BASE as YML
dev = 'cuda'; STEPS = int(os.environ.get('STEPS', '1500')); EVERY = int(os.environ.get('EVERY', '250'))


def to_nf(b):
  r = b.rays
  f = lambda x: x.reshape(-1, x.shape[-1]).contiguous()
  d = f(r.directions); n = d.norm(dim=-1, keepdim=True)
  N = d.shape[0]
  return dict(origin=f(r.origins) * 0.25, direction=d / n, viewdir=f(r.viewdirs), near=f(r.near)[:, 0] * 0.25 * n[:, 0], far=f(r.far)[:, 0] * 0.25 * n[:, 0],
              embed_idx=torch.zeros(N, dtype=torch.int32, device=dev), bg_rgb=torch.ones(N, 3, device=dev), rgb=f(b.rgb))


print(f'# scratch/nerfacto_psnr_modes.py: {STEPS} steps of 16384 fresh rays (64 patches of 16 x 16), yml sizes, validation PSNR on 4096 fixed rays')
val = to_nf(scene_batch(np.random.default_rng(999), 16, 16, dev))
rows = {}
for cdt in ('fp16', 'bf16', 'fp32'):
  for seed in (0, 1):
    model = NerfactoModel(NerfactoConfig(**dict(YML, warmup_steps=50)), compute_dtype=cdt, seed=3 + seed)
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=dev).manual_seed(100 + seed)
    out = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(STEPS):
      batch = to_nf(scene_batch(rng, 64, 16, dev))
      model.train_step(batch, u01=[torch.rand(16384, generator=g, device=dev) for _ in range(3)])
      if (s + 1) % EVERY == 0:
        r = model.render(val, s + 1, chunk_size=4096)
        out.append(float(-10 * torch.log10(((r['rgb'] - val['rgb'])**2).mean())))
    torch.cuda.synchronize()
    rows[(cdt, seed)] = out
    print(f'{cdt} seed {seed}: {(time.perf_counter() - t0) / STEPS * 1e3:.1f} ms/step incl. batch generation; PSNR at steps {list(range(EVERY, STEPS + 1, EVERY))}: {[round(p, 2) for p in out]}', flush=True)
fin = {c: np.mean([rows[(c, s)][-1] for s in (0, 1)]) for c in ('fp16', 'bf16', 'fp32')}
print('final PSNR, mean of 2 seeds:', {k: round(v, 2) for k, v in fin.items()})
