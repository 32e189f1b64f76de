"""Margins of the round-4 parity tests (tests/test_gpu_parity_tight.py): worst per-leaf errors, printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_parity_tight import _step_and_replay
from tests.test_gpu_train_step import SMALL
def show(tag, errs, stats, ostats):
  w = max(errs.items(), key=lambda kv: kv[1][0]); w2 = max(errs.items(), key=lambda kv: kv[1][1])
  print(f'{tag}: loss rel err {abs(float(stats["loss"]) / float(ostats["loss"]) - 1):.2e}; worst max-error / leaf max {w[1][0]:.2e} ({w[0]}); '
        f'worst relative L2 {w2[1][1]:.2e} ({w2[0]}); {len(errs)} leaves', flush=True)
show('fp32, base2, ReLU decisions replayed, no ray masked', *_step_and_replay(list(SMALL), 'fp32'))
gin3 = [g for g in SMALL if not g.startswith('Model.num_') and 'data_loss_type' not in g] + [
    "Model.num_levels = 3", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 32", "Model.raydist_fn = @jnp.reciprocal",
    "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract", "Model.num_glo_features = 4", "Config.data_coarse_loss_mult = 0.1"]
show('fp32, 3 levels contract + reciprocal + GLO 4 + charb', *_step_and_replay(gin3, 'fp32', near=(0.05, 0.3), far=1e6))
ginw = [g for g in SMALL if 'data_loss_type' not in g] + ["Config.transient_type = 'withmask'", "Model.num_glo_features = 48"]
show('fp32, static masks + GLO 48', *_step_and_replay(ginw, 'fp32', n_patch=2))
ginr = [g.replace('patch_size = 8', 'patch_size = 16') for g in SMALL] + ["Config.transient_type = 'robustnerf'", "Config.robustnerf_inlier_quantile = 0.8"]
show('fp32, RobustNeRF 0.8', *_step_and_replay(ginr, 'fp32', n_patch=2, P=16, inlier=0.3))
ginb = [g for g in SMALL if 'net_width' not in g] + ["PropMLP.net_width = 256", "NerfMLP.net_width = 1024"]
errs, stats, ostats = _step_and_replay(ginb, 'bf16', quant=True)
big = {k: v for k, v in errs.items() if v[2] >= 1024}
show('bf16 full width vs bf16-rounded oracle (leaves >= 1024 entries)', big, stats, ostats)
show('bf16 full width vs bf16-rounded oracle (all leaves)', errs, stats, ostats)

# round 5: full width at 256 rays in fp32 with replayed masks; bf16 with the encoder inside the comparison
show('fp32 FULL WIDTH (8x1024 + 4x256), 256 rays, ReLU decisions replayed, no ray masked', *_step_and_replay(ginb, 'fp32', n_patch=4, P=8))
errs, stats, ostats = _step_and_replay(ginb, 'bf16', quant=True, replay_feats=False)
show('bf16 full width, oracle computes its own IPE features (encoder inside; leaves >= 1024 entries)', {k: v for k, v in errs.items() if v[2] >= 1024}, stats, ostats)
show('bf16 full width, oracle computes its own IPE features (all leaves)', errs, stats, ostats)
