"""Gin variants the reference accepts, each through the forward-per-level + whole-step-gradient comparison against the fp32 oracle
(tests/test_gpu_train_step._run_case) -- the permanent form of scratch/config_fuzz2.py (profiles/r05_config_fuzz.txt), which found a
capture bug (a model without a view layer) and a capacity off-by-one this round.  MipNeRF360/internal/models.py:46-72, 359-391;
configs.py:47-136."""
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_train_step import SMALL, _run_case

BOTH = lambda *names: [f'{m}.{n}' for m in ('NerfMLP', 'PropMLP') for n in names]      # (the oracle's ModelCfg has one value for both MLPs)
VARIANTS = {
    'deg_view_2': BOTH('deg_view = 2'),
    'deg_view_6': BOTH('deg_view = 6'),
    'max_deg_16': ["NerfMLP.max_deg_point = 16", "PropMLP.max_deg_point = 16"],
    'max_deg_3': ["NerfMLP.max_deg_point = 3", "PropMLP.max_deg_point = 3"],
    'glo_1': ["Model.num_glo_features = 1"],
    'glo_127': ["Model.num_glo_features = 127"],
    'nerf_depth_2': ["NerfMLP.net_depth = 2"],
    'skip_layer_2': BOTH('skip_layer = 2'),
    'prop_depth_1': ["PropMLP.net_depth = 1"],
    'bottleneck_384': BOTH('bottleneck_width = 384'),
    'nerf_width_384': ["NerfMLP.net_width = 384"],
    'nerf_width_192_depth_4': ["NerfMLP.net_width = 192", "NerfMLP.net_depth = 4"],
    'prop_width_64': ["PropMLP.net_width = 64"],
    'samples_48_24': ["Model.num_prop_samples = 48", "Model.num_nerf_samples = 24"],
    'samples_4_4': ["Model.num_prop_samples = 4", "Model.num_nerf_samples = 4"],
    'samples_256_512': ["Model.num_prop_samples = 256", "Model.num_nerf_samples = 512"],
    'cylinder': ["Model.ray_shape = 'cylinder'"],
    'single_jitter_off': ["Model.single_jitter = False"],
    'near_anneal': ["Model.near_anneal_rate = 0.5"],      # (_run_case steps at train_frac 0.37: init_s_near = 0.26)
    'no_anneal_no_dilation': ["Model.anneal_slope = 0.", "Model.dilation_multiplier = 0.", "Model.dilation_bias = 0."],
    'resample_padding': ["Model.resample_padding = 0.01"],
    'density_bias_0': BOTH('density_bias = 0.'),
    'rgb_padding_0': BOTH('rgb_padding = 0.'),
    'charb_coarse_loss': ["Config.data_loss_type = 'charb'", "Config.data_coarse_loss_mult = 0.3"],
    'interlevel_only': ["Config.distortion_loss_mult = 0.", "Config.interlevel_loss_mult = 0.5"],
    'patch_4': ["Config.patch_size = 4"],
    'glo_levels3_contract': ["Model.num_glo_features = 4", "Model.num_levels = 3", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract",
                             "Model.raydist_fn = @jnp.reciprocal"],
    'view_depth_4': ["NerfMLP.net_depth_viewdirs = 4"],
    'prop_rgb_viewdirs': ["PropMLP.disable_rgb = False", "PropMLP.bottleneck_width = 128", "Config.data_coarse_loss_mult = 0.2"],
    'no_opaque_bg_half': ["Model.opaque_background = False", "Model.bg_intensity_range = (0.5, 0.5)"],
    'no_viewdirs_glo': ["Model.use_viewdirs = False", "Model.num_glo_features = 4"],
    'log_raydist': ["Model.raydist_fn = @jnp.log"],
    'sqrt_raydist_contract': ["Model.raydist_fn = @jnp.sqrt", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract"],
}


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_variant_forward_and_gradients_vs_oracle(name):
  extra = VARIANTS[name]
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys] + extra
  P = 4 if name == 'patch_4' else 8
  _run_case(gin, n_patch=max(1, 64 // (P * P)), P=P)


FT = ["Config.finetune_enable = True", "Config.finetune_lr_init = 0.02", "Config.finetune_lr_delay_steps = 20", "Config.finetune_adam_beta1 = 0.8",
      "Config.finetune_adam_eps = 1e-7", "Model.num_glo_features = 4"]
FINETUNE_VARIANTS = {
    'glo': [],
    'withmask_charb_levels3_contract': ["Config.transient_type = 'withmask'", "Config.data_loss_type = 'charb'", "Model.num_levels = 3",
                                        "Model.num_prop_samples = 64", "Model.num_nerf_samples = 32", "NerfMLP.warp_fn = @coord.contract",
                                        "PropMLP.warp_fn = @coord.contract", "Model.raydist_fn = @jnp.reciprocal"],
    'robustnerf': ["Config.transient_type = 'robustnerf'", "Config.patch_size = 16"],
    'hanerf': ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "NerfMLP.bottleneck_width = 128"],
    'nerfw_glo48': ["Config.transient_type = 'nerfw'", "Model.num_transient_features = 16", "NerfMLP.bottleneck_width = 128",
                    "Model.num_glo_features = 48"],
    'no_viewdirs_weight_decay': ["Model.use_viewdirs = False", "Config.weight_decay_mults = {'NerfMLP_0': 0.1}"],
    'coarse_loss_prop_rgb': ["PropMLP.disable_rgb = False", "PropMLP.bottleneck_width = 128", "Config.data_coarse_loss_mult = 0.2"],
}


@pytest.mark.parametrize('name', sorted(FINETUNE_VARIANTS))
def test_finetune_step_forward_and_gradients_vs_oracle(name):
  """The finetune stage's step (train.py:97-109; create_train_step(model, config, True) + create_finetune_optimizer) of each model
  family against the oracle, whose finetune form tests/test_oracle_vs_reference_model.py holds to the reference executed: the plain
  data loss whatever transient_type is (robustnerf thresholds ignored, no mask / uncertainty terms), every leaf's gradient, and an
  update that moves the embedding tables only, by the finetune_* schedule and Adam knobs."""
  extra = FT + FINETUNE_VARIANTS[name]
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys]
  for e in extra:         # (a later binding of the same key wins, as in gin)
    gin = [g for g in gin if g.split('=')[0].strip() != e.split('=')[0].strip()] + [e]
  P = 16 if name == 'robustnerf' else 8
  _run_case(gin, n_patch=max(1, 64 // (P * P)), P=P, inlier=0.3 if name == 'robustnerf' else None, finetune=True)


@pytest.mark.parametrize('name,err', [('nerf_depth_5', NotImplementedError), ('rawnerf', AssertionError)])
def test_refusals(name, err):
  """A skip concat after the LAST trunk layer is not built (models.py:451-456 would create it for net_depth = 5, 9); a data loss other
  than 'mse' / 'charb' asserts as the reference's compute_data_loss does (train_utils.py:96-101)."""
  extra = {'nerf_depth_5': ["NerfMLP.net_depth = 5"], 'rawnerf': ["Config.data_loss_type = 'rawnerf'"]}[name]
  keys = {e.split('=')[0].strip() for e in extra}
  with pytest.raises(err):
    _run_case([g for g in SMALL if g.split('=')[0].strip() not in keys] + extra)
