"""Phase timeline of the persistent NT kernel (trace build)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib
dev = 'cuda'
M, N, K = int(os.environ.get('M', 131072)), 1024, int(os.environ.get('K', 1024))
DATA = os.environ.get('DATA', 'randn')      # randn | zeros | relu (post-relu activations: half zeros)
if DATA == 'zeros':
  A = torch.zeros(M, K, device=dev).bfloat16(); Bt = torch.zeros(N, K, device=dev).bfloat16()
else:
  A = torch.randn(M, K, device=dev); A = (A.clamp(min=0) if DATA == 'relu' else A).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 32).bfloat16()
print('operands:', DATA)
bias = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
call = lambda: _lib.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N)
cd = _lib.lib().cdll
g, it = int(os.environ.get('SG', 0)), int(os.environ.get('SI', 0))
if g: cd.hugs_debug_set_stagger(g, it)
for _ in range(20): call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
print(f'SG={g} SI={it}: {e0.elapsed_time(e1)/20*1e3:.1f} us per launch')
tr = torch.zeros(256 * 16 * 4, dtype=torch.int64, device=dev)
cd = _lib.lib().cdll
cd.hugs_debug_set_trace.argtypes = [ctypes.c_void_p]
cd.hugs_debug_set_trace(tr.data_ptr())
call(); torch.cuda.synchronize()
cd.hugs_debug_set_trace(None)
t = tr.cpu().numpy().reshape(256, 16, 4)[:, :8].astype(np.float64)
t0 = t[:, 0, 0].min()
print('first4 iters  mean', (t[:, :, 1] - t[:, :, 0]).mean(0).round())
print('rest of loop  mean', (t[:, :, 2] - t[:, :, 1]).mean(0).round())
print('epilogue      mean', (t[:, :, 3] - t[:, :, 2]).mean(0).round())
print('tile total    mean', (t[:, 1:, 0] - t[:, :-1, 0]).mean(0).round())
print('WG0 starts', (t[0, :, 0] - t0).round(), 'kernel span', t[:, -1, 3].max() - t0)
