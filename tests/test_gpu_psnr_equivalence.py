"""Matched-PSNR guard at the BENCHMARKED network (NerfMLP 8x1024 + PropMLP 4x256, 1024 rays x (64+128) samples per step):
bf16 (the benchmarked mode) and fp32 (parity mode) trained from the same init / batches / jitter reach the same validation
PSNR.  Evidence under profiles/: r03_psnr_seeds_bf16_vs_fp32.txt (3 seeds x 1000 steps of a 2000-step schedule, i.e. still
at a high learning rate: final gap +0.47 +- 0.58 dB with fp32's own seed spread at 0.52 dB and +-2 dB excursions of single
evaluations) and r03_psnr_short_schedule.txt (3 seeds x a complete 400-step schedule: every evaluation of every seed within
0.08 dB).  The guard runs the complete short schedule once."""
import pytest

pytestmark = pytest.mark.gpu


def test_bf16_training_tracks_fp32_at_the_benchmarked_network():
  from tests.analytic_scene import psnr_run
  a = psnr_run('bf16', 400, seed=0, max_steps=400, every=50)
  b = psnr_run('fp32', 400, seed=0, max_steps=400, every=50)
  pa, pb = [p for _, p, _ in a], [p for _, p, _ in b]
  assert pa[-1] > pa[0] + 6.0 and pb[-1] > pb[0] + 6.0, (pa, pb)          # both learn (8.4 -> 16.8 dB in the recorded runs)
  assert max(abs(x - y) for x, y in zip(pa, pb)) < 0.5, (pa, pb)          # every evaluation (recorded: <= 0.08 dB)
  assert abs(pa[-1] - pb[-1]) < 0.3, (pa, pb)
