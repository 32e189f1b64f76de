import sys, os, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nerf_hugs_amd.internal import configs, train_utils, models
configs.clear_config(); configs.parse_config_files_and_bindings(None, bench.GIN)
config = configs.make_config()
model, state, render_fn, _, _ = train_utils.setup_model(config, 1, compute_dtype='bf16')
b = bench.synth_batch(256, 16, 0, 'cuda')     # 65536 rays
rays = b.rays.map(lambda x: x.reshape(256, 256, -1))
fn = functools.partial(render_fn, state.params, 1.0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = models.render_image(fn, rays, None, config, verbose=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f'render_image 256x256 (65536 rays, chunk {config.render_chunk_size}): {dt*1e3:.1f} ms  {65536/dt/1e3:.1f} k rays/s  fwd MFMA frac {65536/dt*2.2617e9/2.5e15:.3f}')
