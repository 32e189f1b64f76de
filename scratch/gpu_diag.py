import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
from oracle import torch_ref as R
from nerf_hugs_amd.internal import models as M
config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(SMALL)
batch = H.synth_rays(1, 8, 5)
N = 64; L = 2
gen = torch.Generator(device='cuda').manual_seed(11)
u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(L)]
gen.manual_seed(11)
orays = H.oracle_rays(batch)
s32, g32, r32, h32 = R.loss_and_grad(cfg, oparams, orays, batch.rgb.reshape(-1, 3), 0.37, [u.cpu() for u in u01])
dbl = lambda t: t.double() if t.is_floating_point() else t
op64 = {'params': {m: {k: ({kk: vv.double() for kk, vv in v.items()} if isinstance(v, dict) else v.double()) for k, v in sub.items()} for m, sub in oparams['params'].items()}}
s64, g64, r64, h64 = R.loss_and_grad(cfg, op64, {k: dbl(v) for k, v in orays.items()}, batch.rgb.reshape(-1, 3).double(), 0.37, [u.cpu() for u in u01])
state, stats, gen = train_step(gen, state, batch, 0.37, None)
torch.cuda.synchronize()
grad = model.engine('cuda').ws.get('grad', (model.layout.size + 64,))
print('loss', float(stats['loss']), float(s32['loss']), float(s64['loss']))
for lf in model.layout.leaves:
    name = '/'.join(lf['path'])
    g = model.layout.view(grad, lf['path']).cpu().double()
    ref = g64[name]; sc = ref.abs().max().clamp(min=1e-30)
    print(f'{name:32s} |g|max {float(sc):.2e}  hip-vs-f64 {float((g-ref).abs().max()/sc):.2e}  orc32-vs-f64 {float((g32[name].double()-ref).abs().max()/sc):.2e}')
