"""Golden vectors for the NERFACTO model / field / loss WIRING, recorded by importing and EXECUTING the reference's own
/root/reference/nerfacto/models/nerfacto.py (`Model`, `Loss`: forward_rays, NerfactoField, HashMLPDensityField,
ImplicitMask, every compute_*_loss) on CPU in float32 (`enable_amp=False`, `enable_tcnn_mlp=False` -- the form every
shipped yml selects).  The only thing replaced is the third-party `tinycudann` package (un-vendored, not installable
here): `tests/golden/_tcnn_standin.py` backs `tcnn.Encoding` with oracle/hashgrid_ref.py, so the hash-grid / SH
arithmetic itself stays PARITY UNPINNED while everything the reference wires around it is pinned.
Build container only; only data is committed (tests/golden/ref_nerfacto_model.npz).

    python tests/golden/gen_nerfacto_model_fixtures.py

Per case: rays, parameters (in the build's naming: prop{i}/{table,w0,b0,w1,b1}, field/{table,w0,b0,w1,b1,c0,cb0,..},
appearance, transient, mask/{table,m0,mb0,..}; Linear weights as [fan_in, fan_out]), the torch.rand draws the sampler
consumed, Model.forward outputs, Loss.forward terms, and d loss / d parameter of every leaf (NaN-filled where the
reference leaves `.grad` None: proposal networks on steps without a proposal update)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/nerfacto'

SMALL = dict(num_levels=6, base_res=16, max_res=256, log2_hashmap_size=12, features_per_level=2, hidden_dim=32, geo_feat_dim=15,
             hidden_dim_color=32, enable_tcnn_mlp=False, num_embedding=40, num_nerf_samples_per_ray=8,
             num_proposal_samples_per_ray=(32, 16), num_proposal_iterations=2,
             proposal_net_args_list=[dict(hidden_dim=16, log2_hashmap_size=10, num_levels=3, max_res=64, base_res=16, features_per_level=2),
                                     dict(hidden_dim=16, log2_hashmap_size=11, num_levels=4, max_res=128, base_res=16, features_per_level=2)],
             proposal_weights_anneal_max_num_iters=1000)

CASES = {
    'base': dict(cfg=dict(SMALL, proposal_initial_sampler='uniform'), N=64, step=300, contraction=False),
    # a step on which the proposal networks get no update (interval 2 at step 2501): their .grad stays None
    'base_noprop': dict(cfg=dict(SMALL, proposal_initial_sampler='uniform', rgb_loss_type='charb'), N=64, step=2501, contraction=False),
    'withmask': dict(cfg=dict(SMALL, proposal_initial_sampler='piecewise', transient_type='withmask', withmask_transient_weight=0.25,
                              use_appearance_embedding=True, appearance_embedding_dim=8, opaque_background=True,
                              rgb_loss_type='charb', distortion_loss_mult=0.001, proposal_histogram_padding=0.005),
                     N=64, step=700, contraction=True),
    'robustnerf': dict(cfg=dict(SMALL, proposal_initial_sampler='uniform', transient_type='robustnerf',
                                use_appearance_embedding=True, appearance_embedding_dim=8, opaque_background=True),
                       N=256, step=40, contraction=False),
    'hanerf': dict(cfg=dict(SMALL, proposal_initial_sampler='uniform', transient_type='hanerf', use_appearance_embedding=True,
                            appearance_embedding_dim=8, use_transient_embedding=True, transient_embedding_dim=12,
                            opaque_background=True, num_levels_implicit=4, base_res_implicit=16, max_res_implicit=128,
                            log2_hashmap_size_implicit=10, features_per_level_implicit=2, hidden_dim_implicit=32),
                   N=64, step=120, contraction=False),
}


def export_params(model, grads=False):
  """Reference module tree -> the build's flat naming.  Linear weights transposed to [fan_in, fan_out]."""
  out = {}

  def val(p):
    if grads:
      return np.full(tuple(p.shape), np.nan, np.float32) if p.grad is None else p.grad.detach().numpy().copy()
    return p.detach().numpy().copy()

  def lin(seq, idxs, names, prefix):
    for i, (w, b) in zip(idxs, names):
      out[f'{prefix}/{w}'] = val(seq[i].weight).T.copy()
      out[f'{prefix}/{b}'] = val(seq[i].bias)
  for i, net in enumerate(model.proposal_networks):
    enc = net.mlp_base[0]
    out[f'prop{i}/table'] = val(enc.params).reshape(enc.spec['n_entries'], enc.spec['F'])
    lin(net.mlp_base, (1, 3), (('w0', 'b0'), ('w1', 'b1')), f'prop{i}')
  f = model.field
  enc = f.mlp_base[0]
  out['field/table'] = val(enc.params).reshape(enc.spec['n_entries'], enc.spec['F'])
  lin(f.mlp_base, (1, 3), (('w0', 'b0'), ('w1', 'b1')), 'field')
  lin(f.mlp_head, (0, 2, 4), (('c0', 'cb0'), ('c1', 'cb1'), ('c2', 'cb2')), 'field')
  if model.embedding_appearance is not None:
    out['appearance'] = val(model.embedding_appearance.weight)
  if model.embedding_transient is not None:
    out['transient'] = val(model.embedding_transient.weight)
  if model.implicit_mask is not None:
    enc = model.implicit_mask.grid_encoder
    out['mask/table'] = val(enc.params).reshape(enc.spec['n_entries'], enc.spec['F'])
    lin(model.implicit_mask.mlp_base, (0, 2, 4), (('m0', 'mb0'), ('m1', 'mb1'), ('m2', 'mb2')), 'mask')
  return out


def main():
  if not os.path.isdir(REF):
    raise SystemExit('needs the reference checkout at ' + REF)
  sys.path.insert(0, ROOT)
  sys.path.insert(0, REF)
  sys.path.insert(0, HERE)
  import warnings
  warnings.filterwarnings('ignore')
  import _tcnn_standin
  _tcnn_standin.install()
  import importlib.util       # models/__init__.py pulls further model files: load nerfacto.py by path
  import types
  pkg = types.ModuleType('models'); pkg.__path__ = [os.path.join(REF, 'models')]
  sys.modules['models'] = pkg
  spec = importlib.util.spec_from_file_location('models.nerfacto', os.path.join(REF, 'models', 'nerfacto.py'))
  nf = importlib.util.module_from_spec(spec)
  sys.modules['models.nerfacto'] = nf
  spec.loader.exec_module(nf)
  out, real_rand = {}, torch.rand
  T = lambda a: np.asarray(a.detach().numpy())
  for name, case in CASES.items():
    torch.manual_seed(1234 + len(name))
    g = torch.Generator().manual_seed(77 + len(name))
    mc = nf.ModelConfig(**case['cfg'])
    model = nf.Model(mc, bound=2.0, enable_amp=False, enable_scene_contraction=case['contraction'])
    crit = nf.Loss(model)
    # tables at a size that makes the grid features matter (tiny-cuda-nn's +-1e-4 init would leave every field output
    # at its bias), biases away from zero
    with torch.no_grad():
      for mod in model.modules():
        if isinstance(mod, _tcnn_standin.Encoding) and hasattr(mod, 'params'):
          mod.params.copy_((torch.rand(mod.params.shape, generator=g) * 2 - 1) * 0.5)
    N, step = case['N'], case['step']
    d = torch.randn(N, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
    rays = dict(origin=(torch.rand(N, 3, generator=g) - 0.5) * 1.2, direction=d * (0.8 + 0.4 * torch.rand(N, 1, generator=g)), viewdir=d,
                near=0.05 + 0.1 * torch.rand(N, 1, generator=g), far=2.5 + torch.rand(N, 1, generator=g),
                embed_idx=torch.randint(0, mc.num_embedding, (N, 1), generator=g), bg_rgb=torch.rand(N, 3, generator=g),
                rgb=torch.rand(N, 3, generator=g), static_mask=(torch.rand(N, 1, generator=g) < 0.7).float(),
                coord=torch.rand(N, 2, generator=g))
    if name == 'robustnerf':       # whole 16x16 patches; make some pixels clear outliers
      rays['rgb'][:40] = 1.0 - rays['rgb'][:40] * 0.1
    for k, v in rays.items():
      out[f'{name}/rays/{k}'] = T(v)
    for k, v in export_params(model).items():
      out[f'{name}/params/{k}'] = v
    out[f'{name}/spec'] = np.array(json.dumps(dict(cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in case['cfg'].items()},
                                                   N=N, step=step, contraction=case['contraction'])))
    draws = []

    def rec_rand(*a, **k):
      r = real_rand(*a, **{kk: vv for kk, vv in k.items() if kk != 'device'}, generator=g)
      draws.append(r)
      return r
    model.train()
    torch.rand = rec_rand
    try:
      outputs = model(batch=rays, curr_step=step, perturb=True)
    finally:
      torch.rand = real_rand
    for i, r in enumerate(draws):
      out[f'{name}/u01/{i}'] = T(r)
    P = 16 if name == 'robustnerf' else 8
    data_shape = (N // (P * P), P, P)
    extra = {'curr_step': step, 'curr_frac': step / 25000}
    loss, info, extra = crit(outputs=outputs, batch=rays, data_shape=data_shape, is_finetune=False, extra_infos=extra)
    model.zero_grad()
    loss.backward()
    for k, v in outputs.items():
      if isinstance(v, list):
        for i, t in enumerate(v):
          out[f'{name}/out/{k}/{i}'] = T(t)
      else:
        out[f'{name}/out/{k}'] = T(v)
    out[f'{name}/loss'] = np.float64(loss.detach())
    for k, v in info.items():
      out[f'{name}/info/{k}'] = np.float64(v)
    for k, v in export_params(model, grads=True).items():
      out[f'{name}/grads/{k}'] = v
    if name == 'robustnerf':       # second evaluation with the fed-back threshold (extra_infos carries it, train.py:171,201-204)
      out[f'{name}/next_thr'] = np.float64(extra['inlier_threshold'])
      loss2, info2, _ = crit(outputs={k: (v.detach() if torch.is_tensor(v) else [t.detach() for t in v]) for k, v in outputs.items()},
                             batch=rays, data_shape=data_shape, is_finetune=False, extra_infos=extra)
      out[f'{name}/loss_fedback'] = np.float64(loss2)
      for k, v in info2.items():
        out[f'{name}/info_fedback/{k}'] = np.float64(v)
    # finetune-stage loss (Loss.forward is_finetune=True: plain data loss whatever the transient type)
    lossf, infof, _ = crit(outputs={k: (v.detach() if torch.is_tensor(v) else [t.detach() for t in v]) for k, v in outputs.items()},
                           batch=rays, data_shape=data_shape, is_finetune=True, extra_infos=dict(extra))
    out[f'{name}/loss_finetune'] = np.float64(lossf)
    # eval mode: perturb=False, chunked, embedding per eval_embedding ('average' default)
    model.eval()
    with torch.no_grad():
      ev = model(batch={k: v for k, v in rays.items()}, curr_step=step, perturb=False, chunk_size=48)
    for k in ('rgb', 'depth', 'accumulation', 'implicit_mask'):
      if k in ev:
        out[f'{name}/eval/{k}'] = T(ev[k])
  # The nerfacto NeRF-W branch: constructs, then dies on its first forward (nerfacto.py:394-401 format `output_type`, a name
  # never bound in the file).  Recorded so that "not built" is a statement about the reference, not about this build.
  mc = nf.ModelConfig(**dict(SMALL, proposal_initial_sampler='uniform', transient_type='nerfw', use_transient_embedding=True,
                             transient_embedding_dim=12))
  model = nf.Model(mc, bound=2.0, enable_amp=False, enable_scene_contraction=False)
  model.train()
  N = 16
  d = torch.randn(N, 3); d = d / d.norm(dim=-1, keepdim=True)
  try:
    model(batch=dict(origin=torch.rand(N, 3) - 0.5, direction=d, viewdir=d, near=torch.full((N, 1), 0.1), far=torch.full((N, 1), 3.),
                     embed_idx=torch.zeros(N, 1, dtype=torch.long), bg_rgb=torch.ones(N, 3)), curr_step=10, perturb=True)
    out['nerfw/error_type'], out['nerfw/error'] = np.array('none'), np.array('')
  except Exception as e:      # noqa: BLE001
    out['nerfw/error_type'], out['nerfw/error'] = np.array(type(e).__name__), np.array(str(e))
  np.savez_compressed(os.path.join(HERE, 'ref_nerfacto_model.npz'), **out)
  print('wrote ref_nerfacto_model.npz', len(out), 'arrays')


if __name__ == '__main__':
  main()
