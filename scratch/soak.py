"""Soak: thousands of captured train steps with fresh batch tensors and stats read only now and then -- device memory, host RSS, the pinned
stat-slot pool and the loss must stay put.   python scratch/soak.py [steps] [variant: base|hanerf|robust]"""
import sys, os, resource, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerf_hugs_amd.internal import configs, train_utils, random as hr
from tests import hugs_testlib as H

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
variant = sys.argv[2] if len(sys.argv) > 2 else 'base'
gin = list(bench.GIN)
if variant == 'hanerf':
  gin += ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "Model.num_glo_features = 4", "Config.distortion_loss_mult = 0.001"]
if variant == 'robust':
  gin += ["Config.transient_type = 'robustnerf'", "Config.robustnerf_inlier_quantile = 0.8", "Model.num_glo_features = 4"]
configs.clear_config(); configs.parse_config_files_and_bindings(None, gin + ["Config.randomized = True"])
config = configs.make_config()
model, state, render_fn, train_step, lr_fn = train_utils.setup_model(config, 0, compute_dtype='bf16')
pool = [H.synth_rays(4, 16, 100 + i) for i in range(8)]
key = hr.PRNGKey(5)
rss = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
t0 = time.perf_counter()
first = None
for i in range(steps):
  b = pool[i % 8]
  b = b.__class__(rays=b.rays.map(lambda x: x.clone()), rgb=b.rgb.clone())      # fresh tensors every step
  state, stats, key = train_step(key, state, b, min(1.0, i / steps), None)
  if i % 500 == 0 or i == steps - 1:
    loss = float(stats['loss'])
    torch.cuda.synchronize()
    gc.collect()
    if i == 500:
      first = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    print(f'step {i:5d} loss {loss:.5f} psnr {float(stats["psnr"]):6.2f} cuda alloc {torch.cuda.memory_allocated() / 2**20:9.1f} MiB reserved {torch.cuda.memory_reserved() / 2**20:9.1f} MiB '
          f'max rss {rss():8.1f} MiB graph {train_step.graph_active()} {time.perf_counter() - t0:6.1f} s', flush=True)
last = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
print('device memory grew after step 500:', [(b - a) / 2**20 for a, b in zip(first, last)], 'MiB')
