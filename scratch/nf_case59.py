import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_nerfacto import SMALL, _check_nerfacto_vs_oracle
PN = dict(hidden_dim=8, log2_hashmap_size=9, num_levels=3, max_res=32)
kw = dict(SMALL, proposal_initial_sampler='piecewise', rgb_loss_type='charb', opaque_background=False, density_activation='softplus',
          num_proposal_iterations=3, num_proposal_samples_per_ray=(32, 16, 16), proposal_net_args_list=[PN] * 3, use_proposal_weight_anneal=False)
drop = sys.argv[1:]
for d in drop:
  kw.pop(d, None)
for seed in (5, 6, 7, 8, 9, 10):
  try:
    _check_nerfacto_vs_oracle(kw, f'seed {seed}', ray_seed=seed)
    print('ray seed', seed, 'ok', flush=True)
  except AssertionError as e:
    print('ray seed', seed, 'MISMATCH', str(e).strip()[:260].replace('\n', ' '), flush=True)
