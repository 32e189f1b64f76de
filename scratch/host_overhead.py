"""Host-side enqueue time of one train step (no synchronisation inside the loop) vs the GPU time."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
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
for trial in range(3):
  t0 = time.perf_counter()
  for _ in range(20):
    state, stats, rng = train_step(rng, state, batch, 0.5, None)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, wall {1e3 * (t2 - t0) / 20:.2f} ms/step')
# from an idle queue: 2 steps (~300 launches) never fill the HIP queue, so this is pure host cost
for trial in range(5):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(2):
    state, stats, rng = train_step(rng, state, batch, 0.5, None)
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'idle-queue enqueue {1e3 * (t1 - t0) / 2:.2f} ms/step, wall {1e3 * (t2 - t0) / 2:.2f} ms/step')
