"""Is a 60-step training run reproducible to better than tens of percent in the loss AT ALL?  The oracle (CPU, fp32) against ITSELF from
initial weights that differ by one float32 rounding (every weight x (1 +- 6e-8)), same batches, same jitter -- next to the HIP path's run
(tests/test_gpu_trajectory_vs_oracle.py's loop).   python scratch/traj_chaos.py [variant] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_ref as R
from tests.test_gpu_trajectory_vs_oracle import VARIANTS, SCHED, _tree
from tests import hugs_testlib as H

variant = sys.argv[1] if len(sys.argv) > 1 else 'base'
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
gin = [g for g in VARIANTS[variant] if g.split('=')[0].strip() not in {s.split('=')[0].strip() for s in SCHED}] + SCHED
config, model, state, _, train_step, cfg, oparams = H.make_pair(gin, compute_dtype='fp32')
batches = [H.synth_rays(1, 8, 40 + i) for i in range(4)]
N, L = 64, model.num_levels
gen = torch.Generator(device='cuda').manual_seed(77)
names = [n for n, _ in R.flat_leaves(oparams['params'])]
g2 = torch.Generator().manual_seed(1)
runs = []
for tag in ('oracle', 'oracle, weights moved by one rounding'):
  p = {n: t.clone() for n, t in R.flat_leaves(oparams['params'])}
  if tag != 'oracle':
    p = {n: t * (1 + 6e-8 * (2 * torch.randint(0, 2, t.shape, generator=g2) - 1).float()) for n, t in p.items()}
  runs.append(dict(tag=tag, p=p, m={n: torch.zeros_like(p[n]) for n in names}, v={n: torch.zeros_like(p[n]) for n in names}, loss=[], psnr=[]))
hip, hip_psnr = [], []
for i in range(nsteps):
  b = batches[i % 4]
  frac = i / 200.0
  u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(L)]
  state, stats, _ = train_step(u01, state, b, frac, None)
  hip.append(float(stats['loss'])); hip_psnr.append(float(stats['psnr']))
  for r in runs:
    ostats, ograds, _, _ = R.loss_and_grad(cfg, _tree(r['p']), H.oracle_rays(b), b.rgb.reshape(-1, 3), frac, [u.cpu() for u in u01])
    r['loss'].append(float(ostats['loss'])); r['psnr'].append(float(R.mse_to_psnr(ostats['mses'].detach())[-1]))
    r['p'], r['m'], r['v'] = R.adam_update(cfg, r['p'], R.clip_gradients(cfg, ograds), r['m'], r['v'], i)
a, b_ = np.array(runs[0]['loss']), np.array(runs[1]['loss'])
h = np.array(hip)
print(f'== {variant}, {nsteps} steps')
print('step   oracle      oracle(1 ulp)  hip         |oracle(1ulp)/oracle-1|  |hip/oracle-1|')
for i in list(range(0, nsteps, 5)) + [nsteps - 1]:
  print(f'{i:4d}   {a[i]:.6f}    {b_[i]:.6f}       {h[i]:.6f}    {abs(b_[i] / a[i] - 1):.2e}                {abs(h[i] / a[i] - 1):.2e}')
last = slice(nsteps - 8, nsteps)
print(f'mean PSNR of the last 8 steps: oracle {np.mean(runs[0]["psnr"][last]):.2f}  oracle(1 ulp) {np.mean(runs[1]["psnr"][last]):.2f}  hip {np.mean(hip_psnr[last]):.2f} dB')
print(f'mean loss of the last 8 steps: oracle {a[last].mean():.5f}  oracle(1 ulp) {b_[last].mean():.5f}  hip {h[last].mean():.5f}')
