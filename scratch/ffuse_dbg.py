import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import importlib
t = importlib.import_module('test_gpu_nerfacto')
model, batch, u01 = t._cfg5_model_and_batch('bf16', 512)
got = {}
for mode in ('0', '1'):
  os.environ['HUGS_NF_FIELD_FUSE'] = mode
  lv = model.forward(batch, 300, u01=u01, training=True)
  torch.cuda.synchronize()
  st = lv[-1]
  got[mode] = {k: st[k].clone().float() for k in ('Y0', 'Xh', 'H0', 'H1', 'rgb', 'density')}
a, b = got['0'], got['1']
for k in a:
  d = (a[k] - b[k]).abs()
  rows = d.reshape(d.shape[0], -1).max(dim=1).values
  top = torch.topk(rows, 5)
  print(k, 'max', float(d.max()), 'rows>1e-3:', int((rows > 1e-3).sum()), 'top rows', top.indices.tolist(), [round(v, 5) for v in top.values.tolist()])
r = int(torch.topk((a['rgb'] - b['rgb']).abs().max(dim=1).values, 1).indices)
print('row', r, 'rgb', a['rgb'][r].tolist(), b['rgb'][r].tolist())
print('H1 diff in row', float((a['H1'][r] - b['H1'][r]).abs().max()), 'H0', float((a['H0'][r] - b['H0'][r]).abs().max()), 'Xh', float((a['Xh'][r] - b['Xh'][r]).abs().max()), 'Y0', float((a['Y0'][r] - b['Y0'][r]).abs().max()))
# recompute rgb from the fused H1 with torch
c2 = model.lay.view(model.flat, 'field/c2').float(); cb2 = model.lay.view(model.flat, 'field/cb2').float()
ref = torch.sigmoid(b['H1'] @ c2[:256] + cb2 + model.cfg.rgb_bias)
print('fused rgb vs torch on fused H1', float((ref - b['rgb']).abs().max()), 'unfused', float((torch.sigmoid(a['H1'] @ c2[:256] + cb2 + model.cfg.rgb_bias) - a['rgb']).abs().max()))
d = (ref - b['rgb']).abs()
bad = (d > 1e-4)
print('bad per channel', bad.sum(dim=0).tolist())
rows = torch.nonzero(bad.any(dim=1)).reshape(-1)
tiles = rows // 64
print('bad rows', rows.numel(), 'in tiles', torch.unique(tiles).numel(), 'first-pass tiles (t<512):', int((tiles < 512).sum()), 'second:', int((tiles >= 512).sum()))
print('row-in-tile histogram (i block of 16):', torch.bincount((rows % 64) // 16, minlength=4).tolist())
print('r16 histogram:', torch.bincount(rows % 16, minlength=16).tolist())
ut, cnt = torch.unique(tiles, return_counts=True)
print('tiles', ut[:20].tolist(), cnt[:20].tolist())
# error pattern: which k would explain it?  delta a = logit(fused) - logit(ref) for channel 0
la = torch.logit(b['rgb'][rows, 0].double().clamp(1e-6, 1-1e-6)) - torch.logit(ref[rows, 0].double().clamp(1e-6, 1-1e-6))
H = b['H1'][rows].double()
# least squares: la = H @ dw  (dw = corruption of c2[:,0])
sol = torch.linalg.lstsq(H, la[:, None]).solution.reshape(-1)
top = torch.topk(sol.abs(), 6)
print('implied weight corruption at k =', top.indices.tolist(), [round(float(sol[i]), 4) for i in top.indices], 'c2[k,0] =', [round(float(c2[i, 0]), 4) for i in top.indices])
