import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
from oracle import cstepfun as C, torch_ref as R
dev='cuda'
rng = np.random.default_rng(0)
N=256
for variant in ['plain','pow4','zeros','anneal','jit']:
    n_prev, ns, dil = 64, 128, 0.0103125
    t = np.sort(rng.uniform(0, 1, (N, n_prev + 1)).astype(np.float32), -1)
    w = rng.uniform(0, 1, (N, n_prev)).astype(np.float32)
    if variant=='pow4': w = w**4
    if variant=='zeros': w[rng.uniform(size=w.shape) < 0.1] = 0
    w /= w.sum(-1, keepdims=True)
    anneal = 0.7 if variant=='anneal' else 1.0
    ub, mj = R.sample_u_base(ns, variant=='jit')
    jit = (rng.random(N, dtype=np.float32) * np.float32(mj)).astype(np.float32) if variant=='jit' else None
    near = np.full(N, 0.1, np.float32); far = np.full(N, 1.2, np.float32)
    sd_o, td_o, idx_o = C.level_sample(t, w, True, dil, 0., 1., anneal, 0., ub, jit, 0, near, far)
    tdil, wdil = C.max_dilate_weights(t, w, dil, 0., 1.)
    g = lambda a: torch.from_numpy(a).to(dev)
    n_in = 3*n_prev-2
    sd = torch.empty(N, ns + 1, device=dev); td = torch.empty(N, ns + 1, device=dev); idx = torch.empty(N, ns, dtype=torch.int32, device=dev)
    tin = torch.empty(N, n_in+1, device=dev); win = torch.empty(N, n_in, device=dev)
    L.call('hugs_level_sample_fwd', N, g(t), g(w), n_prev, 1, dil, 0., 1., anneal, 0., g(ub), g(jit) if jit is not None else None, 1, ns, 0, g(near), g(far), sd, td, idx, tin, win)
    a = sd.cpu().numpy(); bad = np.argwhere(a.view(np.uint32) != sd_o.view(np.uint32))
    tb = np.argwhere(tin.cpu().numpy().view(np.uint32) != tdil[:,1:-1].view(np.uint32)); wb = np.argwhere(win.cpu().numpy().view(np.uint32) != wdil[:,1:-1].view(np.uint32))
    print(variant, 'sdist nbad', len(bad), bad[:3].tolist(), np.abs(a-sd_o).max(), ' t_in bad', len(tb), ' w_in bad', len(wb), wb[:3].tolist())
    if len(bad):
        r,j = bad[0]; print('   ', a[r,max(j-1,0):j+2], sd_o[r,max(j-1,0):j+2], idx.cpu().numpy()[r,max(j-1,0):j+1], idx_o[r,max(j-1,0):j+1])
