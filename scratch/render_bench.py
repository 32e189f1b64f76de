"""Test-mode rendering throughput: models.render_image (models.py:568-649 of the reference) over a synthetic H x W image at the headline
network (kubric_1024_base.gin nets, 64 + 128 samples), bf16, chunks of Config.render_chunk_size rays.   python scratch/render_bench.py [H] [chunk] [reps]"""
import sys, os, time, functools, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerf_hugs_amd.internal import configs, train_utils, models, utils
from tests import hugs_testlib as H

Hh = int(sys.argv[1]) if len(sys.argv) > 1 else 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
configs.clear_config()
configs.parse_config_files_and_bindings(None, bench.GIN + [f"Config.render_chunk_size = {chunk}"])
config = configs.make_config()
model, state, render_fn, train_step, _ = train_utils.setup_model(config, 0, compute_dtype='bf16')
batch = H.synth_rays(Hh * Hh // 256, 16, 7)
rays = batch.rays.map(lambda x: x.reshape(Hh, Hh, -1).cuda())
fn = functools.partial(render_fn, state.params, 1.0)
out = models.render_image(fn, rays, None, config, verbose=False)      # warm-up (workspace allocation, operand casts)
torch.cuda.synchronize()
ts = []
for _ in range(reps):
  t0 = time.perf_counter()
  out = models.render_image(fn, rays, None, config, verbose=False)
  torch.cuda.synchronize()
  ts.append(time.perf_counter() - t0)
t = min(ts)
n = Hh * Hh
flop_fwd = bench.FLOP_TRAIN_PER_RAY / 3.0          # forward third of the work model
print(json.dumps({"what": "render_image, test mode (compute_extras), bf16", "image": [Hh, Hh], "rays": n, "chunk": chunk, "seconds": round(t, 4),
                  "rays_per_s": round(n / t, 1), "all_reps_s": [round(x, 4) for x in ts], "forward_mfma_frac": round(n / t * flop_fwd / 2.5e15, 4),
                  "rgb_mean": float(out['rgb'].float().mean()), "keys": sorted(k for k in out if not k.startswith('ray_'))}))
