"""hugs_gemm_nt_chain against the per-layer launches: every trunk activation and mask word of the headline network's NerfMLP level, bit for bit;
then the time of the eight trunk layers both ways (HIP events around the forward of the level's MLP)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerf_hugs_amd.internal import configs, train_utils, engine as E, models as M
from tests import hugs_testlib as H

configs.clear_config(); configs.parse_config_files_and_bindings(None, bench.GIN)
config = configs.make_config()
model, state, render_fn, train_step, _ = train_utils.setup_model(config, 0, compute_dtype='bf16')
batch = H.synth_rays(4, 16, 7)
eng = model.engine('cuda')
rays = M.rays_to_dict(batch.rays, 'cuda')
N = 1024
gen = torch.Generator(device='cuda').manual_seed(1)
u01 = [torch.rand(N, generator=gen, device='cuda') for _ in range(2)]
out = {}
for mode in (False, True):
  E._NT_CHAIN = mode
  eng.refresh_weights(state.flat)
  for k, t in eng.ws.bufs.items():
    if torch.is_tensor(t) and k[0].startswith('NerfMLP_0/L1/') and ('Y' in k[0] or 'bits' in k[0]):
      t.zero_()
  levels = eng.forward(state.flat, rays, 0.5, u01, False)
  torch.cuda.synchronize()
  out[mode] = {k[0]: t.clone() for k, t in eng.ws.bufs.items() if torch.is_tensor(t) and k[0].startswith('NerfMLP_0/L1/') and ('/Y' in k[0] or 'bits' in k[0])}
  out[mode]['rgb'] = levels[-1]['rgb_out'].clone()
bad = 0
for k in sorted(out[False]):
  same = torch.equal(out[False][k], out[True][k])
  nz = int((out[True][k] != 0).sum())
  if not same:
    bad += 1
    d = (out[False][k] != out[True][k])
    print(f'{k}: DIFFERENT in {int(d.sum())} of {d.numel()} entries (nonzero in chained: {nz})')
print('chained == per-layer, every trunk activation and mask word:', bad == 0, f'({len(out[False])} buffers)')
for mode in (False, True, False, True):
  E._NT_CHAIN = mode
  for _ in range(3):
    eng.forward(state.flat, rays, 0.5, u01, False)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(20):
    eng.forward(state.flat, rays, 0.5, u01, False)
  e1.record(); torch.cuda.synchronize()
  print(f'chain={mode}: forward of both levels {e0.elapsed_time(e1) / 20 * 1e3:.1f} us')
