"""Option variants of the nerfacto model executed by the reference's own nerfacto/models/nerfacto.py (`Model`, `Loss`, torch autograd),
with the machinery of gen_nerfacto_model_fixtures.py (tinycudann stand-in over oracle/hashgrid_ref.py): `density_activation =
'softplus'` (nerfacto.py:36,702-710,910-918), `use_same_proposal_network` (:66,191-199,334), 4 features per level, a non-opaque
background on random background colours, the reciprocal initial sampler with scene contraction, ONE proposal iteration.
Writes tests/golden/ref_nerfacto_variants.npz (data only).

    python tests/golden/gen_nerfacto_variant_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_nerfacto_model_fixtures as G

S = G.SMALL
VARIANTS = {
    'softplus': dict(cfg=dict(S, proposal_initial_sampler='uniform', density_activation='softplus'), N=64, step=300, contraction=False),
    'same_proposal_network': dict(cfg=dict(S, proposal_initial_sampler='uniform', use_same_proposal_network=True,
                                           proposal_net_args_list=[S['proposal_net_args_list'][0]]), N=64, step=300, contraction=False),
    'features_per_level_4': dict(cfg=dict(S, proposal_initial_sampler='uniform', features_per_level=4), N=64, step=300, contraction=False),
    'not_opaque_charb': dict(cfg=dict(S, proposal_initial_sampler='uniform', opaque_background=False, rgb_loss_type='charb',
                                      use_appearance_embedding=True, appearance_embedding_dim=8), N=64, step=500, contraction=False),
    'reciprocal_contraction': dict(cfg=dict(S, proposal_initial_sampler='reciprocal'), N=64, step=300, contraction=True),
    'one_proposal_iteration': dict(cfg=dict(S, proposal_initial_sampler='uniform', num_proposal_iterations=1, num_proposal_samples_per_ray=(32,),
                                            proposal_net_args_list=[S['proposal_net_args_list'][0]]), N=64, step=300, contraction=False),
}


def main():
  G.CASES = VARIANTS
  real = np.savez_compressed

  def save(path, **out):      # (the main generator's NeRF-W probe is not part of this file)
    real(os.path.join(HERE, 'ref_nerfacto_variants.npz'), **{k: v for k, v in out.items() if not k.startswith('nerfw/')})
  np.savez_compressed = save
  try:
    G.main()
  finally:
    np.savez_compressed = real


if __name__ == '__main__':
  main()
