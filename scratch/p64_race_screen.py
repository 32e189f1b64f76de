"""Race screen of k_gemm_nt_bf16_p64 (the guide: a read placed one barrier too early passes refchecks whenever the DMA happens to land
first): many launches on fresh operands, each compared BITWISE with k_gemm_nt_bf16_pers (HUGS_NT_K64=0), with and without an HBM-bound
kernel running beside them on another stream (it perturbs when the DMAs land), three trunk shapes, forward (mask bits out) and dX (bits in)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_hugs_amd import _lib as L
dev = 'cuda'
ITERS = int(os.environ.get('ITERS', 400))
side = torch.cuda.Stream()
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GiB: the neighbour's traffic
bad = 0
for (M, N, K1, K2) in ((131072, 1024, 1024, 0), (66560, 1024, 1024, 512), (66560, 256, 512, 0)):
  K = K1 + K2
  g = torch.Generator(device=dev).manual_seed(M + K)
  Bt = (torch.randn(N, K, device=dev, generator=g) / K**0.5).bfloat16()
  bias = torch.randn(N, device=dev, generator=g)
  for it in range(ITERS):
    A1 = torch.randn(M, K1, device=dev, generator=g).clamp(min=0).bfloat16()
    A2 = torch.randn(M, K2, device=dev, generator=g).bfloat16() if K2 else None
    res = {}
    for mode in ('0', '1'):
      os.environ['HUGS_NT_K64'] = mode
      y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bits = torch.empty(M * N // 32, dtype=torch.int32, device=dev)
      o = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
      if mode == '1' and it % 2:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
          big.mul_(1.0001)
      L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, bias, 1, None, None, y, N, bits, None)
      L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, None, 0, None, None, o, N, None, bits)
      res[mode] = (y, bits, o)
    torch.cuda.synchronize()
    for a, b in zip(res['0'], res['1']):
      if not torch.equal(a, b):
        bad += 1
        print(f'MISMATCH shape {(M, N, K1, K2)} iteration {it}: {int((a != b).sum())} elements')
  print(f'shape {(M, N, K1, K2)}: {ITERS} iterations x (forward + dX), mismatches so far {bad}')
print('race screen:', 'CLEAN' if bad == 0 else f'{bad} MISMATCHES')
