"""Two nerfacto train steps under config variants; report failures."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
dev = 'cuda'
SMALL = dict(num_levels=4, max_res=64, log2_hashmap_size=10, hidden_dim=16, geo_feat_dim=7, hidden_dim_color=16,
             num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, opaque_background=True,
             use_appearance_embedding=True, appearance_embedding_dim=5, num_embedding=4, distortion_loss_mult=0.01,
             proposal_net_args_list=[dict(hidden_dim=8, log2_hashmap_size=9, num_levels=3, max_res=32)])
V = {
  'base': {},
  'features_per_level 4': dict(features_per_level=4),
  'hidden 256 / color 256': dict(hidden_dim=256, hidden_dim_color=256),
  'hidden 200 / color 130': dict(hidden_dim=200, hidden_dim_color=130),
  'geo 31': dict(geo_feat_dim=31),
  'geo 15 app 48': dict(geo_feat_dim=15, appearance_embedding_dim=48, num_embedding=100),
  'no appearance': dict(use_appearance_embedding=False),
  'levels 16 log2 14': dict(num_levels=16, log2_hashmap_size=14, max_res=512),
  'one proposal iteration': dict(num_proposal_iterations=1, num_proposal_samples_per_ray=(32,)),
  'three proposal iterations': dict(num_proposal_iterations=3, num_proposal_samples_per_ray=(32, 16, 16)),
  'samples 64/32/16': dict(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=16),
  'samples 1024/512/256': dict(num_proposal_samples_per_ray=(1024, 512), num_nerf_samples_per_ray=256),
  'contract + piecewise': dict(enable_scene_contraction=True, proposal_initial_sampler='piecewise'),
  'reciprocal sampler': dict(proposal_initial_sampler='reciprocal'),
  'charb': dict(rgb_loss_type='charb'),
  'no interlevel': dict(interlevel_loss_mult=0.),
  'prop hidden 64 levels 5': dict(proposal_net_args_list=[dict(hidden_dim=64, log2_hashmap_size=12, num_levels=5, max_res=64)]),
  'prop hidden 128 (gemm path)': dict(proposal_net_args_list=[dict(hidden_dim=128, log2_hashmap_size=12, num_levels=5, max_res=64)]),
  'two different prop nets': dict(use_same_proposal_network=False, proposal_net_args_list=[dict(hidden_dim=16, log2_hashmap_size=10, num_levels=3, max_res=32), dict(hidden_dim=32, log2_hashmap_size=11, num_levels=4, max_res=64)]),
  'not opaque': dict(opaque_background=False),
  'softplus density': dict(density_activation='softplus'),
}
for cdt in ('fp32', 'fp16'):
  for name, kw in V.items():
    try:
      cfg = NerfactoConfig(**dict(SMALL, **kw))
      model = NerfactoModel(cfg, compute_dtype=cdt, seed=1)
      N = 128
      g = torch.Generator(device=dev).manual_seed(0)
      d = torch.randn(N, 3, generator=g, device=dev); d = d / d.norm(dim=-1, keepdim=True)
      b = dict(origin=torch.randn(N, 3, generator=g, device=dev) * 0.3, direction=d, viewdir=d, near=torch.full((N,), 0.05, device=dev),
               far=torch.full((N,), 3.0, device=dev), embed_idx=torch.randint(0, 4, (N,), generator=g, device=dev).int(),
               bg_rgb=torch.ones(N, 3, device=dev), rgb=torch.rand(N, 3, generator=g, device=dev))
      L = cfg.num_proposal_iterations + 1
      ls = []
      for _ in range(2):
        res = model.train_step(b, u01=[torch.rand(N, generator=g, device=dev) for _ in range(L)])
        ls.append(float(res['stats'][1]))
      ok = all(np.isfinite(ls)) and bool(torch.isfinite(model.flat).all())
      print(f'{cdt} {name:32s} {"ok" if ok else "NON-FINITE"} {ls[0]:.5f} -> {ls[1]:.5f}', flush=True)
    except Exception as e:
      print(f'{cdt} {name:32s} {type(e).__name__}: {str(e)[:150]}', flush=True)
