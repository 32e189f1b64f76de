"""scratch: hugs_level_sample_fwd in the two summation orders (1 = reference order: numpy-pairwise sums + sequential cumsum;
0 = wave order) at the cfg2 level shapes, 1024 and 8192 rays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd.internal import stepfun
dev = 'cuda'
def t(fn, n=200):
  for _ in range(20): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for N in (1024, 8192):
  near, far = torch.full((N,), 0.1, device=dev), torch.full((N,), 1.2, device=dev)
  u = torch.rand(N, device=dev)
  t0 = torch.tensor([[0., 1.]], device=dev).repeat(N, 1); w0 = torch.ones(N, 1, device=dev)
  sd0, _ = stepfun.level_sample(t0, w0, False, 0., (0., 1.), 0.9, 0., 64, u, None, near, far)
  w1 = torch.rand(N, 64, device=dev) ** 3; w1 /= w1.sum(-1, keepdim=True)
  for name, args in (('level 0: 1 bin -> 64 samples', (t0, w0, False, 0., (0., 1.), 0.9, 0., 64, u, None, near, far)),
                     ('level 1: 64 bins dilated to 190 -> 128 samples', (sd0, w1, True, 0.0103125, (0., 1.), 0.9, 0., 128, u, None, near, far))):
    r = {o: t(lambda: stepfun.level_sample(*args, sum_order=o)) for o in (1, 0)}
    print(f'N={N} {name}: reference order {r[1]:.1f} us, wave order {r[0]:.1f} us')
