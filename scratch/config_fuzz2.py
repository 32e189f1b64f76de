"""Forward + whole-step gradients against the oracle (tests/test_gpu_train_step._run_case) under gin variants the reference accepts,
each also replayed from the hipGraph (jax key, 5 steps) against its eager twin.  Reports pass / the assertion / the refusal."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import hugs_testlib as H
from tests.test_gpu_train_step import SMALL, _run_case
from tests.test_gpu_step_graph import _run as graph_run
import ast, re
_src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config_fuzz.py')).read()
V0 = ast.literal_eval(_src[_src.index('VARIANTS = {') + len('VARIANTS = '):_src.index('}\nfor name, extra') + 1])
VARIANTS = dict(V0)
# (the oracle's ModelCfg has ONE skip_layer / deg_view / density_bias / rgb_padding / bottleneck for both MLPs: set both)
for k_, names in {'deg_view 2': ['deg_view = 2'], 'deg_view 6': ['deg_view = 6'], 'skip_layer 2': ['skip_layer = 2'], 'density softplus bias 0': ['density_bias = 0.'],
                  'rgb_padding 0': ['rgb_padding = 0.'], 'bottleneck 128': ['bottleneck_width = 128'], 'bottleneck 384': ['bottleneck_width = 384']}.items():
  VARIANTS[k_] = [f'{m}.{n}' for m in ('NerfMLP', 'PropMLP') for n in names]
VARIANTS.update({
  'glo 4 + levels 3 + contract': ["Model.num_glo_features = 4", "Model.num_levels = 3", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract", "Model.raydist_fn = @jnp.reciprocal"],
  'view depth 2 + glo 8': ["NerfMLP.net_depth_viewdirs = 2", "Model.num_glo_features = 8"],
  'prop with rgb + viewdirs': ["PropMLP.disable_rgb = False", "PropMLP.bottleneck_width = 128", "Config.data_coarse_loss_mult = 0.2"],
  'no opaque bg + bg 0.5': ["Model.opaque_background = False", "Model.bg_intensity_range = (0.5, 0.5)"],
  'withmask + glo': ["Config.transient_type = 'withmask'", "Model.num_glo_features = 4"],
  'samples 1024/1024': ["Model.num_prop_samples = 1024", "Model.num_nerf_samples = 1024"],
  'samples 256/512': ["Model.num_prop_samples = 256", "Model.num_nerf_samples = 512"],
  'samples 4/4': ["Model.num_prop_samples = 4", "Model.num_nerf_samples = 4"],
  'levels 7': ["Model.num_levels = 7", "Model.num_prop_samples = 16", "Model.num_nerf_samples = 16"],
  'no viewdirs + glo 4': ["Model.use_viewdirs = False", "Model.num_glo_features = 4"],
  'view depth 4': ["NerfMLP.net_depth_viewdirs = 4"],
  'hanerf': ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "NerfMLP.bottleneck_width = 128"],
  'nerfw': ["Config.transient_type = 'nerfw'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "NerfMLP.bottleneck_width = 128"],
  'robustnerf patch 16': ["Config.transient_type = 'robustnerf'", "Config.patch_size = 16"],
  'log raydist': ["Model.raydist_fn = @jnp.log"],
  'sqrt raydist + contract': ["Model.raydist_fn = @jnp.sqrt", "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract"],
})
only = sys.argv[1:] 
for name, extra in VARIANTS.items():
  if only and not any(o in name for o in only):
    continue
  keys = {e.split('=')[0].strip() for e in extra}
  gin = [g for g in SMALL if g.split('=')[0].strip() not in keys] + extra
  P = 4 if 'patch 4' in name else 16 if 'patch 16' in name else 8
  res = []
  try:
    _run_case(gin, n_patch=max(1, 64 // (P * P)), P=P)
    res.append('oracle ok')
  except NotImplementedError as e:
    res.append(f'refused: {str(e)[:80]}')
  except AssertionError as e:
    res.append(f'ORACLE MISMATCH: {str(e)[:160]}')
  except Exception as e:
    res.append(f'{type(e).__name__}: {str(e)[:120]}')
  if res[-1] == 'oracle ok':
    try:
      kw = dict(n_patch=max(2, 128 // (P * P)), P=P)
      e_ = graph_run('0', gin, 5, 'key', **kw); g_ = graph_run('1', gin, 5, 'key', **kw)
      if not g_[6]:
        res.append('graph: not engaged')
      else:
        sc = float(e_[0].abs().max())
        d = float((e_[0] - g_[0]).abs().max())
        res.append('graph == eager' if d == 0 else f'graph vs eager max diff {d:.2e} (scale {sc:.2e})')
    except Exception as e:
      res.append(f'graph {type(e).__name__}: {str(e)[:120]}')
  print(f'{name:45s} ' + ' | '.join(res), flush=True)
