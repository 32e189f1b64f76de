"""view-layer / bottleneck weight-gradient shapes: hugs_gemm_tn over split counts, stand-alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
for (M, Kc, N) in ((131072, 256, 128), (131072, 1024, 256), (524288, 256, 128)):
  X = torch.randn(M, Kc, device=dev).bfloat16(); G = torch.randn(M, N, device=dev).bfloat16()
  dW = torch.empty(Kc, N, device=dev); db = torch.empty(N, device=dev)
  for ns in (16, 32, 64, 128, 256, 512):
    if M % (ns * 64): continue
    ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, N, ns) // 4, device=dev)
    for _ in range(3): L.call('hugs_gemm_tn', 1, M, Kc, N, ns, X, Kc, G, N, dW, db, ws)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): L.call('hugs_gemm_tn', 1, M, Kc, N, ns, X, Kc, G, N, dW, db, ws)
    torch.cuda.synchronize()
    print(f'M={M} Kc={Kc} N={N} nsplit={ns}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us')
