"""Stand-alone timing of hugs_gemm_nt / hugs_gemm_tn over K and M (fixed-cost vs per-K-cost model)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib
dev = 'cuda'
def t_nt(M, N, K, mask=False, reps=20):
  A = torch.randn(M, K, device=dev).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 32).bfloat16()
  bias = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  Y = torch.randn(M, N, device=dev).bfloat16() if mask else None
  if mask:
    call = lambda: _lib.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, None, None, 1, 0, 0, Y, N, None, None, out, N)
  else:
    call = lambda: _lib.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N)
  for _ in range(10): call()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): call()
  e1.record(); torch.cuda.synchronize()
  dt = e0.elapsed_time(e1) / reps * 1e-3
  return dt * 1e6, 2.0 * M * N * K / dt / 1e12
def t_tn(M, Kc, N, ns, reps=20):
  X = torch.randn(M, Kc, device=dev).bfloat16(); G = torch.randn(M, N, device=dev).bfloat16()
  dW = torch.empty(Kc, N, device=dev); db = torch.empty(N, device=dev)
  nb = _lib.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, N, ns)
  slab = torch.empty(nb // 4, device=dev)
  call = lambda: _lib.call('hugs_gemm_tn', 1, M, Kc, N, ns, X, Kc, G, N, dW, db, slab)
  for _ in range(10): call()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): call()
  e1.record(); torch.cuda.synchronize()
  dt = e0.elapsed_time(e1) / reps * 1e-3
  return dt * 1e6, 2.0 * M * N * Kc / dt / 1e12
for M in (131072, 524288):
  for K in (256, 512, 1024, 1536, 2048):
    us, tf = t_nt(M, 1024, K)
    print(f'NT  M={M} N=1024 K={K}: {us:8.1f} us {tf:7.0f} TF')
  us, tf = t_nt(M, 1024, 1024, mask=True)
  print(f'NTm M={M} N=1024 K=1024: {us:8.1f} us {tf:7.0f} TF')
for N in (256, 512):
  us, tf = t_nt(131072, N, 1024); print(f'NT  M=131072 N={N} K=1024: {us:8.1f} us {tf:7.0f} TF')
us, tf = t_nt(65536, 256, 256); print(f'NT  M=65536 N=256 K=256: {us:8.1f} us {tf:7.0f} TF')
us, tf = t_nt(65536, 256, 512); print(f'NT  M=65536 N=256 K=512: {us:8.1f} us {tf:7.0f} TF')
for ns in (4, 8, 16, 32):
  us, tf = t_tn(131072, 1024, 1024, ns); print(f'TN  M=131072 1024x1024 nsplit={ns}: {us:8.1f} us {tf:7.0f} TF (incl. slab reduce)')
us, tf = t_tn(65536, 256, 256, 64); print(f'TN  M=65536 256x256 nsplit=64: {us:8.1f} us {tf:7.0f} TF')
