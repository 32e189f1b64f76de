"""Diagnostic: per-leaf <grad, v> of the HIP step vs the oracle's float64 autograd on the reference fixture."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import ref_model_fixture as FX
from tests.test_gpu_vs_reference_model import _build
from oracle import torch_ref as R
case = sys.argv[1] if len(sys.argv) > 1 else 'hanerf'
config, model, state, train_step, batch, _ = _build(case)
L = model.num_levels
thr = FX.get(case, 'inlier_thresholds') if config.transient_type == 'robustnerf' else None
state, stats, _ = train_step([u.cuda() for u in FX.u01(case, L)], state, batch, float(FX.get(case, 'train_frac')), thr)
torch.cuda.synchronize()
grad = model.engine('cuda').ws.get('grad', (model.layout.size + 64,))
g = {'/'.join(lf['path']): model.layout.view(grad, lf['path']).double().cpu().numpy() for lf in model.layout.leaves}
cfg = FX.oracle_cfg(case); dt = torch.float64
P, rays = FX.param_tree(case, dt), FX.rays_flat(case, dt)
gt = torch.from_numpy(FX.get(case, 'rgb').reshape(-1, 3).astype(np.float64))
othr = None if thr is None else [torch.from_numpy(t.astype(np.float64)) for t in thr]
ov = []
from nerf_hugs_amd.internal import models as M
eng = model.engine('cuda')
if os.environ.get('OWN_POS'):
  model.load_variables(state.flat, FX.param_tree(case)); eng.refresh_weights(state.flat)
  lv = eng.forward(state.flat, M.rays_to_dict(batch.rays, 'cuda'), float(FX.get(case, 'train_frac')), [u.cuda() for u in FX.u01(case, L)], False)
  for l in range(L):
    ov.append((lv[l]['sdist'].double().cpu(), lv[l]['tdist'].double().cpu()))
else:
 for l in range(L):
  sd = torch.from_numpy(FX.get(case, f'train/l{l}_sdist').astype(np.float64))
  ov.append((sd, R.s_to_t(sd, rays['near'], rays['far'], cfg.raydist_fn)))
orig = R.model_forward
R.model_forward = lambda *a, **k: orig(*a, **dict(k, override_samples=ov))
ostats, og, _, _ = R.loss_and_grad(cfg, P, rays, gt, float(FX.get(case, 'train_frac')), FX.u01(case, L), othr)
for i in range(3):
  v = FX.seeded_tree(case, 1000 + i)
  tot_m = tot_o = 0
  rows = []
  for k in sorted(v):
    m = float((g[k] * v[k]).sum()); o = float((og[k].numpy() * v[k]).sum())
    tot_m += m; tot_o += o
    rows.append((abs(m - o), k, m, o))
  print('dir', i, 'hip', tot_m, 'oracle64', tot_o, 'fd', float(FX.get(case, f'fd/dir{i}')))
  for e, k, m, o in sorted(rows, reverse=True)[:6]:
    gm = np.abs(og[k].numpy()).max()
    print(f'   {k:40s} hip {m:+.6f} oracle {o:+.6f}  max|g-go|/max|go| {np.abs(g[k]-og[k].numpy()).max()/max(gm,1e-30):.2e}')
