"""The finetune stage (train.py:97-109 -> train_utils.py:599-605 setup_finetune_model) executed: the reference's own
create_finetune_optimizer (:515-552, Adam on the 'embedding' leaves, everything else frozen, the finetune_* schedule / Adam knobs)
and create_train_step(model, config, True) (:422-444: plain data loss whatever transient_type says, no interlevel / distortion /
weight-decay terms), run under the stand-ins with the machinery of gen_model_fixtures.py on a NeRF-W and on a HA-NeRF model.
`optax.multi_transform` / `set_to_zero` are restated in the stand-in from their documentation (third party); what is pinned is the
reference's partition, loss selection, clip and hyper-parameter routing.  Recorded per case: every loss term / stat of the step, the
clip + update on the seeded synthetic gradient tree, the forward of every level.  Writes tests/golden/ref_model_finetune.npz.

    python tests/golden/gen_model_finetune_fixtures.py        # ~1 min
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_fixtures as G

FT = {'finetune_enable': True, 'finetune_max_steps': 800, 'finetune_lr_init': 0.02, 'finetune_lr_final': 0.002,
      'finetune_lr_delay_steps': 50, 'finetune_lr_delay_mult': 0.1, 'finetune_adam_beta1': 0.85, 'finetune_adam_beta2': 0.995,
      'finetune_adam_eps': 1e-7}
CASES = {
    'ft_nerfw': dict(
        Config=dict(G.CASES['nerfw']['Config'], grad_max_norm=0.05, **FT),
        **{k: v for k, v in G.CASES['nerfw'].items() if k != 'Config'}),
    'ft_hanerf': dict(
        Config=dict(G.CASES['hanerf']['Config'], data_loss_type='mse', weight_decay_mults={'NerfMLP_0': 0.1}, **FT),
        **{k: v for k, v in G.CASES['hanerf'].items() if k != 'Config'}),
}


def main():
  G.CASES = CASES
  G.N_DIRS = 0
  G.FINETUNE = True
  real = np.savez_compressed

  def save(path, **out):
    keep = {k: v for k, v in out.items() if k.split('/')[0] in CASES}
    real(os.path.join(HERE, 'ref_model_finetune.npz'), **keep)
  np.savez_compressed = save
  try:
    G.main()
  finally:
    np.savez_compressed = real


if __name__ == '__main__':
  main()
