"""scratch: k_gemm_nt_bf16_pers5 (HUGS_NT_PERS5=1) against the four-slot persistent kernel (=0), bit for bit, via two processes."""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
  import torch
  from nerf_hugs_amd import _lib as L
  dev = 'cuda'
  outs = {}
  for (M, N, K1, K2, bits) in ((131072, 1024, 1024, 0, 0), (131072, 1024, 1024, 0, 1), (131072, 1024, 512, 0, 1), (65536 + 512, 1024, 1024, 512, 1), (4096 * 17, 512, 256, 0, 0)):
    g = torch.Generator(device=dev).manual_seed(M + K1)
    A1 = torch.randn(M, K1, generator=g, device=dev).bfloat16(); A2 = torch.randn(M, K2, generator=g, device=dev).bfloat16() if K2 else None
    Bt = (torch.randn(N, K1 + K2, generator=g, device=dev) / (K1 + K2) ** 0.5).bfloat16(); bias = torch.randn(N, generator=g, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    if bits:
      bw = torch.zeros(M * N // 32, device=dev, dtype=torch.int32)
      L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, 1, None, None, out, N, bw, None)
      outs[f'{M}_{N}_{K1}_{K2}_bits'] = bw.cpu()
    else:
      L.call('hugs_gemm_nt', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, None, 1, 0, 1, None, 0, None, None, out, N)
    torch.cuda.synchronize()
    outs[f'{M}_{N}_{K1}_{K2}_{bits}'] = out.cpu()
    ref = (torch.cat([A1, A2], 1) if K2 else A1)[:2048].double() @ Bt.double().T + bias.double()
    err = float((out[:2048].double() - ref.clamp(min=0)).abs().max())
    print(sys.argv[1], M, N, K1, K2, bits, 'max err vs fp64 (first 2048 rows)', err, flush=True)
  torch.save(outs, sys.argv[2])
else:
  import torch
  for v in ('0', '1'):
    subprocess.check_call([sys.executable, __file__, v, f'/tmp/pers5_{v}.pt'], env=dict(os.environ, HUGS_NT_PERS5=v))
  a, b = torch.load('/tmp/pers5_0.pt'), torch.load('/tmp/pers5_1.pt')
  for k in a:
    eq = torch.equal(a[k], b[k])
    print(k, 'pers5 == pers:', eq, '' if eq else f'differing elements {int((a[k] != b[k]).sum())} of {a[k].numel()}')
