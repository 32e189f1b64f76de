import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
from oracle import torch_ref as R
from nerf_hugs_amd.internal import models as M
for extra in (["Model.num_levels = 4"], ["Model.num_levels = 3"]):
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys] + extra
  config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin)
  batch = H.synth_rays(1, 8, 5)
  N = 64; L = model.num_levels
  gen = torch.Generator(device='cuda').manual_seed(11)
  u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(L)]
  ostats, ograds, orend, ohist = R.loss_and_grad(cfg, oparams, H.oracle_rays(batch), batch.rgb.reshape(-1, 3), 0.37, [u.cpu() for u in u01], None)
  eng = model.engine('cuda'); eng.refresh_weights(state.flat)
  levels = eng.forward(state.flat, M.rays_to_dict(batch.rays, 'cuda'), 0.37, u01, False)
  for l in range(L):
    a, b = levels[l]['sdist'].cpu().double(), ohist[l]['sdist'].double()
    d = (a - b).abs()
    print(extra, 'L', l, 'sdist relerr', H.relerr(levels[l]['sdist'], ohist[l]['sdist']), 'n>1e-5:', int((d > 1e-5).sum()), 'of', d.numel(),
          'weights relerr', H.relerr(levels[l]['weights'], ohist[l]['weights']))
