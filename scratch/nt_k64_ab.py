"""A/B of the persistent NT kernel with 32-wide stages (HUGS_NT_K64=0) vs 64-wide super-stages of whole cache lines (=1): the trunk shape
[131072 x 1024] x [1024 x 1024]^T, forward (bias + relu + mask bits out) and dX (mask bits in), interleaved in one process, HIP events,
cycles per tile from the kernels' own s_memtime account.  Operands: post-relu N(0,1) activations / N(0,1) gradients, weights N(0,1/K)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, N, K = int(os.environ.get('M', 131072)), 1024, 1024
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M, K, device=dev, generator=g).clamp(min=0).bfloat16()
G = torch.randn(M, K, device=dev, generator=g).bfloat16()
Bt = (torch.randn(N, K, device=dev, generator=g) / K**0.5).bfloat16()
bias = torch.randn(N, device=dev, generator=g)
y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); o = torch.empty_like(y)
bits = torch.empty(M * N // 32, dtype=torch.int32, device=dev)
cyc = torch.zeros(64 * 4 * 2, dtype=torch.int64, device=dev)
fwd = lambda: L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, 1, None, None, y, N, bits, None)
dx = lambda: L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, G, K, None, 0, Bt, K, None, 0, None, None, o, N, None, bits)
res = {}
for rnd in range(6):
  for mode in ('0', '1', 'w4'):
    os.environ['HUGS_NT_K64'] = '0' if mode == '0' else '1'
    os.environ['HUGS_NT_W4'] = '1' if mode == 'w4' else '0'
    for name, fn, epi in (('fwd', fwd, 35), ('dx', dx, 16)):
      for _ in range(3): fn()
      cyc.zero_(); torch.cuda.synchronize()
      L.call('hugs_debug_set_nt_cycles', cyc.data_ptr())
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(20): fn()
      e1.record(); torch.cuda.synchronize()
      L.call('hugs_debug_set_nt_cycles', 0)
      c = cyc.cpu().reshape(64, 4, 2)[epi, 2]
      res.setdefault((name, mode), []).append((e0.elapsed_time(e1) / 20 * 1e3, float(c[0]) / max(1, float(c[1]))))
for (name, mode), v in sorted(res.items()):
  us = np.array([a for a, _ in v]); cy = np.array([b for _, b in v])
  tf = 2.0 * M * N * K / (np.median(us) * 1e-6) / 1e12
  print(f'{name:4s} K64={mode}: us median {np.median(us):7.1f} min {us.min():7.1f}   {tf:7.1f} TF   cycles/tile median {np.median(cy):8.0f}  per-cycle {2*256*256*1024/np.median(cy)/4096:.4f}')
