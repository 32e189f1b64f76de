"""cfg5 field weight-gradient shapes (M = 2 M samples, fp16): hugs_gemm_tn stand-alone, checked against a float matmul on a slice."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M = 2097152
for (Kc, N) in ((128, 256), (256, 128), (256, 256)):
  X = (torch.randn(M, Kc, device=dev) * 0.1).half(); G = (torch.randn(M, N, device=dev) * 0.1).half()
  dW = torch.empty(Kc, N, device=dev); db = torch.empty(N, device=dev)
  tiles, target = (Kc // 128) * (N // 128), 768
  if Kc % 256 == 0 and N % 256 == 0: tiles, target = 1, 256
  units = M // 64
  ns = max(1, min(units, (target + tiles - 1) // tiles))
  while units % ns: ns -= 1
  ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, N, ns) // 4, device=dev)
  for _ in range(3): L.call('hugs_gemm_tn', 2, M, Kc, N, ns, X, Kc, G, N, dW, db, ws)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20): L.call('hugs_gemm_tn', 2, M, Kc, N, ns, X, Kc, G, N, dW, db, ws)
  torch.cuda.synchronize()
  us = (time.perf_counter() - t0) / 20 * 1e6
  ref = X.float().T @ G.float()
  err = float((dW - ref).abs().max()) / float(ref.abs().max())
  print(f'Kc={Kc} N={N} nsplit={ns}: {us:.1f} us = {M * (Kc + N) * 2 / us / 1e6:.2f} TB/s algorithmic, rel err {err:.2e}, bias err {float((db - G.float().sum(0)).abs().max()):.2e}')
