"""Does processing the batch in ray chunks (activations resident in the 256 MiB Infinity Cache between layers) speed up
the forward trunk chain?  8 layers [M,1024]x[1024,1024] with all activations kept (separate buffers, as training does)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib
dev = 'cuda'
M, W, L = 131072, 1024, 8
acts = [torch.randn(M, W, device=dev).bfloat16() for _ in range(L + 1)]
Ws = [(torch.randn(W, W, device=dev) / 32).bfloat16() for _ in range(L)]
bias = torch.zeros(W, device=dev)
def chain(nchunk):
  m = M // nchunk
  for c in range(nchunk):
    for l in range(L):
      x = acts[l][c * m:(c + 1) * m]; y = acts[l + 1][c * m:(c + 1) * m]
      _lib.call('hugs_gemm_nt', 1, m, W, W, 0, x, W, None, 0, Ws[l], W, bias, None, 1, 0, 1, None, 0, None, None, y, W)
for nchunk in (1, 2, 4, 8, 1, 2, 4):
  for _ in range(3): chain(nchunk)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5): chain(nchunk)
  e1.record(); torch.cuda.synchronize()
  dt = e0.elapsed_time(e1) / 5 * 1e-3
  print(f'chunks {nchunk}: {dt*1e3:.3f} ms for {L} layers  {2.0*M*W*W*L/dt/1e12:.0f} TF')
