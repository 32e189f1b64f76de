"""fused 256-wide trunk tail (layers 1..3 + density head) vs the launches it replaces, stand-alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
for M in (16384, 65536, 1048576):
  g = torch.Generator(device=dev).manual_seed(1)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  nl = 3
  Y0 = rn(M, 256).clamp_(min=0).bfloat16()
  Wt = [(rn(256, 256) * (2.0 / 256)**0.5).bfloat16() for _ in range(nl)]
  bias = [rn(256) * 0.1 for _ in range(nl)]
  wd, bd = rn(256) * 0.1, rn(1)
  Y = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(nl)]
  bits = [torch.zeros(M * 256 // 32, dtype=torch.int32, device=dev) for _ in range(nl)]
  raw, dens = torch.empty(M, device=dev), torch.empty(M, device=dev)
  ptrs = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
  a_w, a_b, a_y, a_bits = ptrs(Wt), ptrs(bias), ptrs(Y), ptrs(bits)
  def fused():
    L.call('hugs_mlp256_tail_fwd', 1, M, nl, Y0, a_w.ctypes.data, a_b.ctypes.data, a_y.ctypes.data, a_bits.ctypes.data, wd, bd, -1.0, raw, dens)
  def layers():
    x = Y0
    for l in range(nl):
      L.call('hugs_gemm_nt_bits', 1, M, 256, 256, 0, x, 256, None, 0, Wt[l], 256, bias[l], 1, None, None, Y[l], 256, bits[l], None)
      x = Y[l]
    L.call('hugs_density_fwd', 1, M, 256, x, 256, wd, bd, -1.0, raw, dens)
  for name, fn in (('layer by layer', layers), ('fused', fused)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f'M={M}: {name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us')
