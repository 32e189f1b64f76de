"""Phase trace of k_mlp256_chain3_fwd8 (build: EXTRA=-DMC_TRACE scratch/build_variant.sh libmctrace hugs_mlpfuse.hip ... see r5_j.sh)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, nl = 1048576, 3
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
Y0 = rn(M, 256).clamp_(min=0).bfloat16()
Wt = [(rn(256, 256) * (2.0 / 256)**0.5).bfloat16() for _ in range(nl)]
bias = [rn(256) * 0.1 for _ in range(nl)]
wd, bd = rn(256) * 0.1, rn(1)
Y = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(nl)]
bits = [torch.zeros(M * 256 // 32, dtype=torch.int32, device=dev) for _ in range(nl)]
raw, dens = torch.empty(M, device=dev), torch.empty(M, device=dev)
ptrs = lambda ts: np.ascontiguousarray([t.data_ptr() for t in ts], np.uint64)
a_w, a_b, a_y, a_bits = ptrs(Wt), ptrs(bias), ptrs(Y), ptrs(bits)
f = lambda: L.call('hugs_mlp256_tail_fwd', 1, M, nl, Y0, a_w.ctypes.data, a_b.ctypes.data, a_y.ctypes.data, a_bits.ctypes.data, wd, bd, -1.0, raw, dens)
for _ in range(3): f()
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
L.lib().cdll.hugs_mc_trace_read(buf)
t = np.array(buf, dtype=np.float64).reshape(2, 8, 16)[:, :, :10]
names = ['L1', 'B1', 'bits+dma', 'L2', 'B2', 'bits+L3', 'vmcnt', 'B3', 'tail']
for w in range(2):
  d = np.diff(t[w], axis=1)[1:]      # skip the first tile
  print('wave', 0 if w == 0 else 7, {n: int(v) for n, v in zip(names, d.mean(0))}, 'tile', int((t[w, 2:, 0] - t[w, 1:-1, 0]).mean()))
