"""Phase timeline of k_gemm_nt_bf16_big<4> (scratch/libhugs_trace.so, built with -DHUGS_TRACE).
Run with HUGS_LIB_PATH=scratch/libhugs_trace.so."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib
dev = 'cuda'
M, N, K = int(os.environ.get('M', 131072)), 1024, int(os.environ.get('K', 1024))
A = torch.randn(M, K, device=dev).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 32).bfloat16()
bias = torch.zeros(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
call = lambda: _lib.call('hugs_gemm_nt', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N)
cd0 = _lib.lib().cdll
if os.environ.get('STAG'):
  g_, it_ = map(int, os.environ['STAG'].split(','))
  cd0.hugs_debug_set_stagger(g_, it_)
for _ in range(20): call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): call()
e1.record(); torch.cuda.synchronize()
print('kernel time us', e0.elapsed_time(e1) / 20 * 1e3, 'TF', 2.0 * M * N * K / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e12)
nwg = (M // 256) * (N // 256)
tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
cd = _lib.lib().cdll
cd.hugs_debug_set_trace.argtypes = [ctypes.c_void_p]
assert cd.hugs_debug_set_trace(tr.data_ptr()) == 0
call(); torch.cuda.synchronize()
cd.hugs_debug_set_trace(None)
t = tr.cpu().numpy().reshape(nwg, 8)
t0 = t[:, 0][t[:, 0] > 0].min()
T = (t[:, :5] - t0).astype(np.float64)
hw = t[:, 6]; xcc = t[:, 7] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 16 + cu
print('kernel span (cycles):', T[:, 4].max(), ' tiles', nwg, 'distinct CU ids', len(np.unique(cuid)))
pro, main, epi, ack = T[:, 1] - T[:, 0], T[:, 2] - T[:, 1], T[:, 3] - T[:, 2], T[:, 4] - T[:, 3]
for name, x in (('prologue', pro), ('mainloop', main), ('epi issue', epi), ('store ack', ack), ('total', T[:, 4] - T[:, 0])):
  print(f'{name:10s} mean {x.mean():9.0f}  p10 {np.percentile(x,10):9.0f}  p50 {np.percentile(x,50):9.0f}  p90 {np.percentile(x,90):9.0f}  max {x.max():9.0f}')
# per-CU sequence: gaps between one tile's end and the next tile's start on the same CU
gaps = []
for c in np.unique(cuid):
  idx = np.nonzero(cuid == c)[0]
  idx = idx[np.argsort(T[idx, 0])]
  for a, b in zip(idx[:-1], idx[1:]):
    gaps.append(T[b, 0] - T[a, 4])
gaps = np.array(gaps)
print('inter-tile gap on a CU: mean', gaps.mean(), 'p50', np.percentile(gaps, 50), 'p90', np.percentile(gaps, 90))
# how synchronised are the epilogues? histogram of epilogue start times
h, e = np.histogram(T[:, 2], bins=40)
print('epilogue-start histogram over the kernel (tiles per bin):', h.tolist())
first = np.sort(T[:, 0])
print('first-wave start spread (cycles): p50', np.percentile(first[:256], 50), 'max', first[:256].max())
for c in np.unique(cuid)[:2]:
  idx = np.nonzero(cuid == c)[0]; idx = idx[np.argsort(T[idx, 0])]
  print('CU', c, [[int(v) for v in T[i, :5]] for i in idx])
