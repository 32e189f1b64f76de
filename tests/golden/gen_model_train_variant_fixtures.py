"""Loss / optimizer option variants of the reference's own train_step (MipNeRF360/internal/train_utils.py:386-477, executed under the
stand-ins with the machinery of gen_model_fixtures.py, finite differences switched off): a coarse data loss on a proposal MLP that
renders colour + charb padding + non-default interlevel / distortion weights; `disable_multiscale_loss` + weight decay + both gradient
clips + a learning-rate warm-up and Adam hyper-parameters off their defaults.  Recorded: every loss term / stat of the step, the
clip + update on the seeded synthetic gradient tree, the forward of every level.  Writes tests/golden/ref_model_train_variants.npz.

    python tests/golden/gen_model_train_variant_fixtures.py        # ~1 min
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_model_fixtures as G

PROP_RGB = dict(G.SMALL_PROP, net_width=128, disable_rgb=False, bottleneck_width=128, net_width_viewdirs=128)
CASES = {
    'coarse_charb': dict(
        Config={'data_loss_type': 'charb', 'charb_padding': 0.01, 'data_coarse_loss_mult': 0.3, 'interlevel_loss_mult': 0.5,
                'distortion_loss_mult': 0.05, 'randomized': True, 'patch_size': 4},
        Model=dict(G.BASE_MODEL, num_levels=3, num_prop_samples=32, num_nerf_samples=32, num_glo_features=4),
        NerfMLP=G.SMALL_NERF, PropMLP=PROP_RGB, n_patch=2, P=4, near=0.1, far=1.2, hist_step=1),
    'decay_clips_schedule': dict(
        Config={'data_loss_type': 'mse', 'disable_multiscale_loss': True, 'weight_decay_mults': {'NerfMLP_0': 0.02, 'PropMLP_0/Dense_1': 0.5},
                'grad_max_norm': 0.05, 'grad_max_val': 0.002, 'lr_init': 0.004, 'lr_final': 0.0004, 'lr_delay_steps': 100, 'lr_delay_mult': 0.1,
                'adam_beta1': 0.8, 'adam_beta2': 0.99, 'adam_eps': 1e-8, 'max_steps': 5000, 'distortion_loss_mult': 0.0,
                'randomized': True, 'patch_size': 4},
        Model=dict(G.BASE_MODEL, num_levels=2, num_prop_samples=32, num_nerf_samples=48),
        NerfMLP=G.SMALL_NERF, PropMLP=G.SMALL_PROP, n_patch=2, P=4, near=0.1, far=1.2, hist_step=1),
}


def main():
  G.CASES = CASES
  G.N_DIRS = 0      # (no float64 finite differences: the five main cases carry those)
  real = np.savez_compressed

  def save(path, **out):
    keep = {k: v for k, v in out.items() if k.split('/')[0] in CASES}
    real(os.path.join(HERE, 'ref_model_train_variants.npz'), **keep)
  np.savez_compressed = save
  try:
    G.main()
  finally:
    np.savez_compressed = real


if __name__ == '__main__':
  main()
