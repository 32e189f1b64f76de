"""scratch: stand-alone timing of the trunk-shape TN (dW) GEMM and, next to it, the forward NT GEMM (HUGS_LIB_PATH picks the build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, N, K = 131072, 1024, 1024
A = torch.randn(M, K, device=dev).bfloat16(); G = torch.randn(M, N, device=dev).bfloat16()
Bt = (torch.randn(N, K, device=dev) / 32).bfloat16(); bias = torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
dW = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
ns = int(os.environ.get('NSPLIT', 16))
ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N, ns) // 4, device=dev)
def t(fn, n=20):
  for _ in range(3): fn()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
tn = t(lambda: L.call('hugs_gemm_tn', 1, M, K, N, ns, A, K, G, N, dW, db, ws))
nt = t(lambda: L.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N))
fl = 2.0 * M * N * K
print(f"{os.environ.get('HUGS_LIB_PATH','default').split('/')[-1]:14s} TN {tn:7.1f} us {fl/tn/1e6:7.1f} TF   NT {nt:7.1f} us {fl/nt/1e6:7.1f} TF")
