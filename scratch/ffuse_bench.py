"""Stand-alone timing of hugs_nf_field_fwd (csrc/hugs_fieldfuse.hip) at the cfg5 field shape: M = 2 M samples."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, S = int(os.environ.get('M', 2097152)), 128
N = M // S
dt = 2
tdt = torch.float16
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.1)
X0 = torch.zeros(M, 128, dtype=tdt, device=dev); X0[:, :32] = r(M, 32).to(tdt)
W0t, W1x, C0t, C1t = r(256, 128).to(tdt), r(128, 256).to(tdt), r(256, 128).to(tdt), r(256, 256).to(tdt)
b0, b1, cb0, cb1, c2, cb2 = r(256), r(128), r(256), r(256), r(256, 3), r(4)
tmpl = r(N, 128).to(tdt); sel = torch.ones(M, device=dev)
Y0, raw, Xh = torch.empty(M, 256, dtype=tdt, device=dev), torch.empty(M, dtype=tdt, device=dev), torch.empty(M, 128, dtype=tdt, device=dev)
H0, H1 = torch.empty(M, 256, dtype=tdt, device=dev), torch.empty(M, 256, dtype=tdt, device=dev)
bY0, bH0 = torch.empty(M * 8, dtype=torch.int32, device=dev), torch.empty(M * 8, dtype=torch.int32, device=dev)
dens, rgb = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
def run():
  L.call('hugs_nf_field_fwd', dt, M, S, X0, 128, W0t, 128, W1x, C0t, C1t, b0, b1, cb0, cb1, c2, cb2, tmpl, 64, sel, Y0, raw, Xh, H0, H1, bY0, bH0, dens, rgb, 0, -1.0)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(os.environ.get('HUGS_LIB_PATH', 'default'), 'M', M, 'us', round(e0.elapsed_time(e1) * 100, 1))
ref = torch.sigmoid(H1.float() @ c2 + cb2[:3])
d = (ref - rgb).abs()
print('  rgb vs torch on the kernel H1: max', float(d.max()), 'bad rows per channel', (d > 1e-4).sum(dim=0).tolist())
import ctypes, numpy as np
cd = L.lib().cdll
if hasattr(cd, 'hugs_ff_trace_read'):
  buf = np.zeros(256, np.int64)
  cd.hugs_ff_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
  tr = buf.reshape(2, 8, 16)
  names = ['L0', '-', 'sync', 'L1 mma+copyY0', 'L1 epi', 'sync', 'C0(+loads,copyXh)', '-', 'sync', 'put inputs', 'C1(+copyH0,rgb)', '-', 'sync', 'final']
  if os.environ.get('HUGS_FF_WAVES', '8') != '4':
    names = ['copyH1+L0', 'sync', 'bits+copyY0', 'L1', 'sync', 'C0(+loads,copyXh)', 'sync', 'bits+put', 'C1(+copyH0,rgb)', 'sync', 'final']
  for w in range(2):
    print('wave', 0 if w == 0 else 3, 'phase cycles per tile:')
    for ti in range(1, 6):
      last = 14 if os.environ.get('HUGS_FF_WAVES', '8') == '4' else 11
      d = np.diff(tr[w, ti][:last + 1]); nxt = tr[w, ti + 1, 0] - tr[w, ti, last]
      print('   tile', ti, ' '.join(f'{n}={int(v)}' for n, v in zip(names, d)), 'loop=', int(nxt), 'total', int(tr[w, ti + 1, 0] - tr[w, ti, 0]))
# checksums of every output (A/B of kernel forms: HUGS_FF_WAVES=4 vs default must agree bit for bit except rgb's summation order)
import hashlib
for name, t in (('Y0', Y0), ('raw', raw), ('Xh', Xh), ('H0', H0), ('H1', H1), ('bY0', bY0), ('bH0', bH0), ('dens', dens)):
  print('  ', name, hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:12])
print('   rgb sum', float(rgb.double().sum()))
