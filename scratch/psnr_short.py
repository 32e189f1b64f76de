"""scratch: calibration of the short matched-PSNR guard (tests/test_gpu_psnr_equivalence.py): bf16 vs fp32, 400 steps with the
schedule compressed to 400 steps, evaluations every 50 steps, seeds 0..2."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.analytic_scene import psnr_run
for seed in range(3):
  a = psnr_run('bf16', 400, seed, max_steps=400, every=50)
  b = psnr_run('fp32', 400, seed, max_steps=400, every=50)
  print(seed, 'bf16', [round(p, 2) for _, p, _ in a]); print(seed, 'fp32', [round(p, 2) for _, p, _ in b], flush=True)
