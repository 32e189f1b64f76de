"""Per-half-iteration cycle stamps of k_gemm_nt_bf16_p64 (trace build): H0 (no barrier, no DMA) vs H1 (barrier + 8 LDS-DMAs)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib
dev = 'cuda'
M, N, K = 131072, 1024, 1024
A = torch.randn(M, K, device=dev).clamp(min=0).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 32).bfloat16()
bias = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
call = lambda: _lib.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N)
for _ in range(10): call()
tr = torch.zeros(256 * 16 * 4 + 256 * 64, dtype=torch.int64, device=dev)
cd = _lib.lib().cdll
cd.hugs_debug_set_trace.argtypes = [ctypes.c_void_p]
cd.hugs_debug_set_trace(tr.data_ptr())
call(); torch.cuda.synchronize()
cd.hugs_debug_set_trace(None)
h = tr.cpu().numpy()[256 * 16 * 4:256 * 16 * 4 + 512].reshape(256, 2).astype(np.float64)
n = 8 * 15      # tiles x super-stages 1..15
print('H0 halves (no barrier, no DMA) mean cycles:', (h[:, 0] / n).mean().round(), ' min/max over WGs', (h[:, 0] / n).min().round(), (h[:, 0] / n).max().round())
print('H1 halves (barrier + 8 DMA)    mean cycles:', (h[:, 1] / n).mean().round(), ' min/max over WGs', (h[:, 1] / n).min().round(), (h[:, 1] / n).max().round())
