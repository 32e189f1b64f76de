"""Matched-PSNR evidence at the BENCHMARKED network (NerfMLP 8x1024 + PropMLP 4x256, 1024 rays x (64+128) samples):
bf16 (the benchmarked mode) vs fp32 (parity mode), several seeds (init, batches and jitter all follow the seed),
validation PSNR on a fixed 4096-ray set.  usage: psnr_seeds.py [steps] [seeds]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.analytic_scene import psnr_run

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nseed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows, t0 = {}, time.time()
for seed in range(nseed):
  for dt in ('bf16', 'fp32'):
    t1 = time.time()
    rows[(seed, dt)] = psnr_run(dt, steps, seed)
    print(f'# seed {seed} {dt}: {time.time() - t1:.1f} s', flush=True)
print(f'# validation PSNR (dB), NerfMLP 8x1024 + PropMLP 4x256, 1024 rays x (64+128) per step, {steps} steps, {nseed} seeds; total {time.time() - t0:.0f} s')
print('# step ' + ' '.join(f'  s{s}_bf16  s{s}_fp32   diff' for s in range(nseed)))
for i, (st, _, _) in enumerate(rows[(0, 'bf16')]):
  print(f'{st:6d} ' + ' '.join(f'{rows[(s, "bf16")][i][1]:9.3f} {rows[(s, "fp32")][i][1]:9.3f} {rows[(s, "bf16")][i][1] - rows[(s, "fp32")][i][1]:+7.3f}' for s in range(nseed)))
tail = lambda r: np.mean([p for _, p, _ in r[-3:]])          # mean of the last three evaluations
fin = {k: tail(v) for k, v in rows.items()}
gaps = np.array([fin[(s, 'bf16')] - fin[(s, 'fp32')] for s in range(nseed)])
bf = np.array([fin[(s, 'bf16')] for s in range(nseed)]); fp = np.array([fin[(s, 'fp32')] for s in range(nseed)])
print(f'# final PSNR (mean of the last 3 evaluations): bf16 {bf.mean():.3f} +- {bf.std():.3f}, fp32 {fp.mean():.3f} +- {fp.std():.3f}')
print(f'# gap bf16 - fp32 per seed: {np.round(gaps, 3).tolist()}  mean {gaps.mean():+.3f} dB, std {gaps.std():.3f} dB; seed-to-seed spread of fp32 alone: {fp.max() - fp.min():.3f} dB')
