import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
def check(M, N, K1, K2, relu=1, mask=False, r1=False, rb=False):
    A1 = torch.randn(M, K1, device=dev).bfloat16(); A2 = torch.randn(M, K2, device=dev).bfloat16() if K2 else None
    Bt = (torch.randn(N, K1 + K2, device=dev) / (K1 + K2) ** 0.5).bfloat16(); bias = torch.randn(N, device=dev)
    rbt = torch.randn(M // 64, N, device=dev) if rb else None; mk = torch.randn(M, N, device=dev).bfloat16() if mask else None
    rr = torch.randn(M, device=dev) if r1 else None; rc = torch.randn(N, device=dev) if r1 else None
    outs = []
    for mode in (0, 3):
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        L.call('hugs_gemm_nt_tiles', mode, 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, rbt, 64, N, relu, mk, N, rr, rc, out, N)
        outs.append(out)
    torch.cuda.synchronize()
    print(f'M={M} N={N} K={K1}+{K2} relu={relu} mask={mask} r1={r1} rb={rb}: persistent == big: {torch.equal(outs[0], outs[1])}', flush=True)
check(256, 256, 256, 0); check(1024, 1024, 1024, 512); check(512, 256, 512, 0, relu=0, mask=True, r1=True); check(768 + 256, 256, 256, 0, rb=True)
check(131072, 1024, 1024, 0); check(65536 + 256, 512, 512, 0, relu=0, mask=True)
def perf(M, N, K, mode, mask=False):
    A = torch.randn(M, K, device=dev).bfloat16(); Bt = (torch.randn(N, K, device=dev) / 32).bfloat16(); bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); mk = torch.randn(M, N, device=dev).bfloat16() if mask else None
    f = lambda: L.call('hugs_gemm_nt_tiles', mode, 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 0 if mask else 1, mk, N, None, None, out, N)
    for _ in range(30): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 30 * 1e-3
    print(f'perf M={M} N={N} K={K} mode={mode} mask={mask}: {dt*1e6:.1f} us {2*M*N*K/dt/1e12:.0f} TF', flush=True)
for rep in range(2):
    for mode in (0, 3):
        perf(131072, 1024, 1024, mode); perf(131072, 1024, 1024, mode, mask=True); perf(131072, 1024, 512, mode); perf(65536, 256, 256, mode)
