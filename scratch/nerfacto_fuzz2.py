"""Random COMBINATIONS of the nerfacto options (models/nerfacto.py ModelConfig) through tests/test_gpu_nerfacto._check_nerfacto_vs_oracle:
one train_step in fp32 against oracle.nerfacto_ref -- bins / weights of every level, colour, loss terms, every parameter's gradient.
python scratch/nerfacto_fuzz2.py [seed] [n]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_nerfacto import SMALL, _check_nerfacto_vs_oracle

PN = lambda **k: dict(dict(hidden_dim=8, log2_hashmap_size=9, num_levels=3, max_res=32), **k)
GROUPS = {
    'transient': [None, None, dict(transient_type='withmask', withmask_transient_weight=0.25), dict(transient_type='robustnerf', robustnerf_inlier_quantile=0.7)],
    'contract': [None, dict(enable_scene_contraction=True)],
    'sampler': [None, dict(proposal_initial_sampler='piecewise'), dict(proposal_initial_sampler='reciprocal')],
    'loss': [None, dict(rgb_loss_type='charb'), dict(rgb_loss_type='charb', rgb_charb_loss_padding=0.01)],
    'bg': [None, dict(opaque_background=False)],
    'density': [None, dict(density_activation='softplus')],
    'grid': [None, dict(features_per_level=4), dict(num_levels=6, max_res=96), dict(log2_hashmap_size=12)],
    'field': [None, dict(geo_feat_dim=31), dict(hidden_dim=32, hidden_dim_color=24), dict(hidden_dim=200, hidden_dim_color=130), dict(geo_feat_dim=15, hidden_dim_color=64)],
    'appearance': [None, dict(use_appearance_embedding=False), dict(appearance_embedding_dim=16)],
    'proposal': [None, dict(use_same_proposal_network=True),
                 dict(num_proposal_iterations=1, num_proposal_samples_per_ray=(32,)),
                 dict(num_proposal_iterations=3, num_proposal_samples_per_ray=(32, 16, 16), proposal_net_args_list=[PN()] * 3),
                 dict(proposal_net_args_list=[PN(hidden_dim=16, log2_hashmap_size=10), PN(hidden_dim=32, log2_hashmap_size=11, num_levels=4, max_res=64)]),
                 dict(proposal_net_args_list=[PN(hidden_dim=24, num_levels=9, max_res=48)])],
    'samples': [None, dict(num_nerf_samples_per_ray=16), dict(num_proposal_samples_per_ray=(64, 32), num_nerf_samples_per_ray=24)],
    'reg': [None, dict(distortion_loss_mult=0.1, interlevel_loss_mult=0.5), dict(proposal_histogram_padding=0.05), dict(use_proposal_weight_anneal=False)],
}
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rnd = random.Random(seed)
tally = {}
for i in range(count):
  picks = {g: rnd.choice(o) for g, o in GROUPS.items()}
  kw = dict(SMALL)
  for g, v in picks.items():
    if v:
      if g == 'samples' and 'num_proposal_samples_per_ray' in v and 'num_proposal_iterations' in (picks['proposal'] or {}):
        v = {k: x for k, x in v.items() if k != 'num_proposal_samples_per_ray'}
      kw.update(v)
  if os.environ.get('FUZZ_ONLY') and i not in [int(x) for x in os.environ['FUZZ_ONLY'].split(',')]:
    continue
  tag = '; '.join(f'{k}={v}' for g, d in picks.items() if d for k, v in d.items())
  try:
    _check_nerfacto_vs_oracle(kw, f'draw {i}')
    res = 'ok'
  except NotImplementedError as e:
    res = f'refused: {str(e)[:100]}'
  except AssertionError as e:
    res = f'MISMATCH: {str(e).strip()[:320]}'.replace('\n', ' ')
  except Exception as e:
    res = f'{type(e).__name__}: {str(e)[:160]}'
  k0 = res.split(':')[0].split(' ')[0]
  tally[k0] = tally.get(k0, 0) + 1
  print(f'{i:3d} {res:40s} {tag}', flush=True)
  torch.cuda.empty_cache()
print('tally', tally)
