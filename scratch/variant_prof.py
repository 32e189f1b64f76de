"""scratch: N steps of one transient variant (argv[1] = base | hanerf | nerfw) for a rocprofv3 kernel trace."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from nerf_hugs_amd.internal import configs, train_utils, random as hr
base = B.GIN[:2] + ["Config.distortion_loss_mult = 0.001", "Model.raydist_fn = @jnp.reciprocal", "Model.num_glo_features = 4",
                    "NerfMLP.warp_fn = @coord.contract", "PropMLP.warp_fn = @coord.contract"] + B.GIN[5:]
extra = {'base': [], 'hanerf': ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 128", "Config.data_loss_mult = 0.5"],
         'nerfw': ["Config.transient_type = 'nerfw'", "Model.num_transient_features = 16"]}[sys.argv[1]]
configs.clear_config()
configs.parse_config_files_and_bindings(None, base + extra)
config = configs.make_config(batch_size=1024)
model, state, _, train_step, _ = train_utils.setup_model(config, 0, compute_dtype='bf16')
batch = B.synth_batch(4, 16, 1000, torch.device('cuda'))
batch.rays.near.uniform_(0.05, 0.3); batch.rays.far.fill_(1e6)
rng = hr.PRNGKey(0)
for _ in range(16):
  state, stats, rng = train_step(rng, state, batch, 0.5, None)
torch.cuda.synchronize()
