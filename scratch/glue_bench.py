import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
N, S = 16384, 128; M = N * S
def t(f, n=10):
  for _ in range(3): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): f()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
sh = torch.randn(N, 16, device=dev); Y1 = torch.randn(M, 128, device=dev).half(); app = torch.randn(N, 48, device=dev)
X = torch.empty(M, 128, device=dev, dtype=torch.float16)
for ngeo, napp in ((15, 48), (15, 0), (0, 48), (0, 0)):
  print('head_input ngeo', ngeo, 'napp', napp, f"{t(lambda: L.call('hugs_nf_head_input', M, S, 2, sh, Y1, 128, ngeo, app if napp else None, napp, X, 128)):.1f} us")
print('zero_ fp16 [M,128]', f"{t(lambda: X.zero_()):.1f} us")
print('copy fp16 [M,128]', f"{t(lambda: X.copy_(Y1)):.1f} us")
dd = torch.randn(M, device=dev); sel = torch.ones(M, device=dev); dXh = torch.randn(M, 128, device=dev).half(); G = torch.empty(M, 128, device=dev, dtype=torch.float16)
print('base_grad', f"{t(lambda: L.call('hugs_nf_base_grad', M, 2, Y1, 128, sel, dd, dXh, 128, 16, 15, G, 128, 0, -1.0)):.1f} us")
print('base_grad no dXh', f"{t(lambda: L.call('hugs_nf_base_grad', M, 2, Y1, 128, sel, dd, None, 0, 16, 0, G, 128, 0, -1.0)):.1f} us")
emb = torch.zeros(3500, 48, device=dev); idx = torch.randint(0, 3500, (N,), device=dev).int()
print('app_bwd', f"{t(lambda: L.call('hugs_nf_app_bwd', N, S, 2, dXh, 128, 31, 48, idx, emb)):.1f} us")
Y1 = torch.randn(M, 128, device=dev).half()
print('head_input ngeo 64 napp 48', f"{t(lambda: L.call('hugs_nf_head_input', M, S, 2, sh, Y1, 128, 64, app, 48, X, 128)):.1f} us")
print('base_grad ngeo 64', f"{t(lambda: L.call('hugs_nf_base_grad', M, 2, Y1, 128, sel, dd, dXh, 128, 16, 64, G, 128, 0, -1.0)):.1f} us")
