"""fused PropMLP dX chain (hugs_mlp256_tail_bwd) vs the launches it replaces (rank1_mask + 3 masked dX GEMMs), stand-alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
for M in (8192, 65536, 1048576):
  g = torch.Generator(device=dev).manual_seed(1)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  acts = [rn(M, 256).clamp_(min=0).bfloat16() for _ in range(4)]
  bits = [torch.randint(-2**31, 2**31 - 1, (M * 256 // 32,), device=dev, generator=g, dtype=torch.int64).int() for _ in range(4)]
  Wn = [(rn(256, 256) * (2.0 / 256)**0.5).bfloat16() for _ in range(3)]
  wd, d_raw = rn(256) * 0.1, rn(M) * 0.3
  G = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(4)]
  ptrs = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
  a_wn, a_bits, a_g = ptrs(Wn), ptrs(bits), ptrs(G)
  def fused():
    L.call('hugs_mlp256_tail_bwd', 1, M, 3, d_raw, wd, a_wn.ctypes.data, a_bits.ctypes.data, a_g.ctypes.data)
  def layers():
    L.call('hugs_rank1_mask', 1, M, 256, d_raw, wd, acts[3], 256, G[3], 256)
    for l in (3, 2, 1):
      L.call('hugs_gemm_nt_bits', 1, M, 256, 256, 0, G[l], 256, None, 0, Wn[l - 1], 256, None, 0, None, None, G[l - 1], 256, None, bits[l - 1])
  for name, fn in (('layer by layer', layers), ('fused', fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f'M={M}: bwd {name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us', flush=True)
