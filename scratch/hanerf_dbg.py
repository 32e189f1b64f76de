import sys
sys.path.insert(0, '.')
from tests import test_gpu_train_step as T
base = T.SMALL
A = base + ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "Config.data_loss_mult = 0.5"]
B = base + ["Config.data_loss_mult = 0.5", "Config.data_coarse_loss_mult = 0.1", "PropMLP.disable_rgb = False", "PropMLP.bottleneck_width = 128", "NerfMLP.bottleneck_width = 128"]
C = base + ["Model.num_glo_features = 4", "Config.data_loss_mult = 0.5"]
for name, g in (('hanerf-only', A), ('coarse+proprgb', B), ('glo-only', C)):
  try:
    w = T._run_case(g, n_patch=2)
    print(name, 'OK worst', w)
  except AssertionError as e:
    print(name, 'FAIL', str(e)[:300])
