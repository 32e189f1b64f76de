import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
rng = np.random.default_rng(0)
n = 1<<20
def rnd(lo, hi):
    return (np.exp(rng.uniform(lo, hi, n)) * rng.choice([-1, 1], n)).astype(np.float32)
for (la, ha, lb, hb) in [(-5,5,-5,5), (-80,0,-80,0), (-100,-60,-20,20), (-30,0,-90,-60), (-20,0,-20,0)]:
    a, b = rnd(la, ha), rnd(lb, hb)
    o = torch.empty(4*n, device='cuda')
    L.call('hugs_test_arith', torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), n, o)
    o = o.cpu().numpy().reshape(4, n)
    with np.errstate(all='ignore'):
        ref = [a/b, a*b, a+b, a-b]
    print((la,ha,lb,hb), [int((o[k].view(np.uint32) != ref[k].view(np.uint32)).sum()) for k in range(4)])
    bad = np.argwhere(o[0].view(np.uint32) != ref[0].view(np.uint32))[:3, 0]
    for i in bad: print('   div', a[i], b[i], o[0][i], ref[0][i])
    bad = np.argwhere(o[1].view(np.uint32) != ref[1].view(np.uint32))[:3, 0]
    for i in bad: print('   mul', a[i], b[i], o[1][i], ref[1][i])
