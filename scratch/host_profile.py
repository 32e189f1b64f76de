import sys, cProfile, pstats
sys.path.insert(0, '.')
import torch
import bench as B
from nerf_hugs_amd.internal import configs, train_utils, random as hr
configs.clear_config()
configs.parse_config_files_and_bindings(None, B.GIN)
config = configs.make_config(batch_size=1024)
model, state, render_fn, train_step, lr_fn = train_utils.setup_model(config, 0, compute_dtype='bf16')
batch = B.synth_batch(4, 16, 1000, torch.device('cuda'))
rng = hr.PRNGKey(0)
for _ in range(10):
  state, stats, rng = train_step(rng, state, batch, 0.5, None)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
  state, stats, rng = train_step(rng, state, batch, 0.5, None)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(18)
