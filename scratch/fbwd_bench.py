"""Stand-alone timing of hugs_nf_field_bwd (csrc/hugs_fieldfuse.hip) at the cfg5 field shape: M = 2 M samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, S = int(os.environ.get('M', 2097152)), 128
N = M // S
dt, tdt = 2, torch.float16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.1)
G1 = r(M, 256).to(tdt)
C1n, C0n, W1xn, W0n = r(256, 256).to(tdt), r(128, 256).to(tdt), r(256, 128).to(tdt), r(128, 256).to(tdt)
bH0 = torch.randint(-2**31, 2**31 - 1, (M * 8,), generator=g, device=dev, dtype=torch.int64).int()
bY0 = torch.randint(-2**31, 2**31 - 1, (M * 8,), generator=g, device=dev, dtype=torch.int64).int()
dd, sel, raw = r(M), torch.ones(M, device=dev), r(M).to(tdt)
eidx = torch.randint(0, 100, (N,), generator=g, device=dev).int()
G0, Gb, Gy0, dX0 = torch.empty(M, 256, dtype=tdt, device=dev), torch.empty(M, 128, dtype=tdt, device=dev), torch.empty(M, 256, dtype=tdt, device=dev), torch.zeros(M, 128, dtype=tdt, device=dev)
demb = torch.zeros(100, 48, device=dev)
def run():
  L.call('hugs_nf_field_bwd', dt, M, S, G1, C1n, C0n, W1xn, W0n, bH0, bY0, dd, sel, raw, 64, 48, eidx, G0, Gb, Gy0, dX0, 128, demb, 0, 0, -1.0)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print('field_bwd M', M, 'us', round(e0.elapsed_time(e1) * 100, 1))
