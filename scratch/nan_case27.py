import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL
extra = ["Config.transient_type = 'robustnerf'", "Config.patch_size = 16", "Model.num_glo_features = 4", "Model.num_levels = 3",
         "Model.num_prop_samples = 48", "Model.num_nerf_samples = 24", "Model.raydist_fn = @jnp.reciprocal", "Model.ray_shape = 'cylinder'",
         "Model.use_viewdirs = False", "Model.opaque_background = False", "Model.bg_intensity_range = (0.5, 0.5)", "Config.data_loss_type = 'charb'",
         "NerfMLP.skip_layer = 2", "PropMLP.skip_layer = 2", "NerfMLP.density_bias = 0.", "PropMLP.density_bias = 0.",
         "Config.weight_decay_mults = {'NerfMLP_0': 0.05}"]
drop = set(sys.argv[1:])
extra = [e for e in extra if not any(d in e for d in drop)]
gin = list(SMALL)
for e in extra:
  gin = [g for g in gin if g.split('=')[0].strip() != e.split('=')[0].strip()] + [e]
config, model, state, render_fn, train_step, cfg, oparams = H.make_pair(gin)
rob = any('robustnerf' in e for e in extra)
P = 16 if rob else 8
batch = H.synth_rays(max(1, 64 // (P * P)), P, 5, near=(0.05, 0.3), far=1e6)
gen = torch.Generator(device='cuda').manual_seed(11)
L = model.num_levels
state, stats, gen = train_step(gen, state, batch, 0.37, np.full((L, 1), 0.3, np.float32) if rob else None)
torch.cuda.synchronize()
eng = model.engine('cuda')
grad = eng.ws.get('grad', (model.layout.size + 64,))
for lf in model.layout.leaves:
  g = model.layout.view(grad, lf['path'])
  if not torch.isfinite(g).all():
    bad = ~torch.isfinite(g)
    rows = bad.any(-1).nonzero().flatten().tolist() if g.dim() == 2 else []
    print('/'.join(lf['path']), tuple(g.shape), 'bad', int(bad.sum()), 'rows', rows[:40], 'cols/row', int(bad.sum()) // max(1, len(rows)))
for k, t in eng.ws.bufs.items():
  if not torch.is_tensor(t):
    continue
  if 'X0' in k[0] and t.dim() == 2:
    a = t.float().abs()
    print('ws', k[0], tuple(t.shape), 'max abs', float(a.max()), 'col of max', int(a.max(0).values.argmax()), 'n>1.5:', int((a > 1.5).sum()))
  if t.is_floating_point() and not torch.isfinite(t.float()).all():
    bad = ~torch.isfinite(t.float())
    msg = ''
    if t.dim() == 2:
      msg = f"rows {bad.any(-1).nonzero().flatten()[:10].tolist()} n_rows {int(bad.any(-1).sum())} cols {bad.any(0).nonzero().flatten()[:40].tolist()}"
    print('ws', k, tuple(t.shape), 'non-finite', int(bad.sum()), msg)
print('loss', float(stats['loss']))
