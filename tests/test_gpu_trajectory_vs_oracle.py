"""north_star's "at matched PSNR": the SAME training run on both sides -- same initial weights, same ray batches, same jitter draws, 40
full steps (forward, losses, backward, clip, Adam with its schedule and bias corrections, weight re-cast) through the HIP path in
fp32 mode and through the oracle on the CPU.

What can be asserted about such a pair of runs was measured first (scratch/traj_chaos.py, profiles/r05_training_trajectory_chaos.txt):
the oracle against ITSELF from weights moved by one float32 rounding agrees to 1e-6 in the loss for ~5 steps, to 1e-3 around step 10 and
to only 5-40 % from step 30 on (Adam's normalised updates turn a rounding of a near-zero gradient entry into an O(lr) parameter
difference) -- and the HIP path's distance from the oracle follows the same curve.  So: per-step parity for the first steps (the
per-step tests make that statement at 1e-4 for every option), and for the run as a whole the statistic that "matched PSNR" means --
the loss and PSNR averaged over the last eight steps, within the spread two oracle runs show between themselves."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_gpu_train_step import SMALL, HANERF

SCHED = ["Config.lr_init = 0.003", "Config.lr_final = 0.0003", "Config.lr_delay_steps = 0", "Config.max_steps = 200"]
VARIANTS = {
    'base': list(SMALL),
    'withmask_glo_charb': list(SMALL) + ["Config.transient_type = 'withmask'", "Model.num_glo_features = 4", "Config.data_loss_type = 'charb'"],
    'hanerf': list(HANERF),
}


def _tree(flatd):
  out = {}
  for n, t in flatd.items():
    d = out
    ks = n.split('/')
    for k in ks[:-1]:
      d = d.setdefault(k, {})
    d[ks[-1]] = t
  return {'params': out}


@pytest.mark.parametrize('variant', ['base', 'hanerf'])
def test_forty_step_training_run_matches_the_oracle_run(variant):
  from tests import hugs_testlib as H
  from oracle import torch_ref as R
  nsteps = 40
  gin = [g for g in VARIANTS[variant] if g.split('=')[0].strip() not in {s.split('=')[0].strip() for s in SCHED}] + SCHED
  config, model, state, _, train_step, cfg, oparams = H.make_pair(gin, compute_dtype='fp32')
  batches = [H.synth_rays(1, 8, 40 + i) for i in range(4)]          # four batches, cycled: the run can fit them
  N, L = 64, model.num_levels
  gen = torch.Generator(device='cuda').manual_seed(77)
  names = [n for n, _ in R.flat_leaves(oparams['params'])]
  p = {n: t.clone() for n, t in R.flat_leaves(oparams['params'])}
  m = {n: torch.zeros_like(p[n]) for n in names}
  v = {n: torch.zeros_like(p[n]) for n in names}
  hip, orc, hip_psnr, orc_psnr = [], [], [], []
  for i in range(nsteps):
    b = batches[i % 4]
    frac = i / 200.0
    u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(L)]
    state, stats, _ = train_step(u01, state, b, frac, None)
    hip.append(float(stats['loss'])); hip_psnr.append(float(stats['psnr']))
    ostats, ograds, _, _ = R.loss_and_grad(cfg, _tree(p), H.oracle_rays(b), b.rgb.reshape(-1, 3), frac, [u.cpu() for u in u01])
    orc.append(float(ostats['loss'])); orc_psnr.append(float(R.mse_to_psnr(ostats['mses'].detach())[-1]))
    p, m, v = R.adam_update(cfg, p, R.clip_gradients(cfg, ograds), m, v, i)
  hip, orc = np.array(hip), np.array(orc)
  rel = np.abs(hip / orc - 1)
  print(f'{variant}: loss {orc[0]:.5f} -> last-8 mean {orc[-8:].mean():.5f} (hip {hip[-8:].mean():.5f}); step-wise |hip/oracle - 1|: '
        f'steps 0-3 {rel[:4].max():.1e}, steps 0-9 {rel[:10].max():.1e}, all {rel.max():.1e}; PSNR last-8 mean {np.mean(orc_psnr[-8:]):.2f} dB (hip {np.mean(hip_psnr[-8:]):.2f})')
  assert orc[-8:].mean() < 0.5 * orc[:4].mean(), 'the run must actually train'
  assert rel[:4].max() < 1e-4, rel[:4]                       # per-step parity while roundings have not been amplified yet (measured 1e-5)
  assert rel[:10].max() < 2e-2, rel[:10]                     # (measured 1.4e-3; the oracle against itself one rounding away: 2.1e-3)
  # the run as a whole: two oracle runs one rounding apart differ by up to 25 % in the last-8 mean loss and 0.3 dB here
  assert abs(np.log(hip[-8:].mean() / orc[-8:].mean())) < np.log(1.6), (hip[-8:].mean(), orc[-8:].mean())
  assert abs(np.mean(hip_psnr[-8:]) - np.mean(orc_psnr[-8:])) < 1.5, (np.mean(hip_psnr[-8:]), np.mean(orc_psnr[-8:]))
