"""Training-PSNR trajectory, bf16 (benchmarked mode) vs fp32 (parity mode), same init / rays / jitter, on a
synthetic analytic scene (textured unit sphere on a white background, pinhole cameras on a ring)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd.internal import configs, train_utils, utils

GIN = ["Config.patch_size = 16", "Config.data_loss_type = 'mse'", "Config.distortion_loss_mult = 0.", "Config.max_steps = 2000",
       "Config.lr_delay_steps = 100", "Model.opaque_background = True", "Model.num_levels = 2", "Model.num_prop_samples = 64",
       "Model.num_nerf_samples = 128", "PropMLP.net_depth = 4", "PropMLP.net_width = 256", "PropMLP.disable_rgb = True",
       "NerfMLP.net_depth = 8", "NerfMLP.net_width = 512"]


def scene_batch(rng, n_patch, P, device):
  """Patches of a pinhole camera on a ring of radius 3 looking at the origin; gt = shaded textured sphere."""
  o = np.zeros((n_patch, P, P, 3), np.float32); d = np.zeros_like(o)
  for i in range(n_patch):
    th = rng.uniform(0, 2 * np.pi); c = np.array([3 * np.cos(th), 3 * np.sin(th), rng.uniform(-0.5, 0.5)], np.float32)
    f = -c / np.linalg.norm(c); r = np.cross(f, [0, 0, 1]); r /= np.linalg.norm(r); u = np.cross(r, f)
    x0, y0 = rng.uniform(-0.35, 0.35, 2)
    px = x0 + (np.arange(P) - P / 2) * 0.004; py = y0 + (np.arange(P) - P / 2) * 0.004
    X, Y = np.meshgrid(px, py)
    dir_ = f[None, None] + X[..., None] * r + Y[..., None] * u
    o[i] = c; d[i] = dir_
  v = d / np.linalg.norm(d, axis=-1, keepdims=True)
  b = (o * v).sum(-1); cc = (o * o).sum(-1) - 1.0; disc = b * b - cc
  hit = disc > 0
  t = -b - np.sqrt(np.maximum(disc, 0))
  p = o + v * t[..., None]
  tex = 0.5 + 0.5 * np.stack([np.sin(6 * p[..., 0]), np.sin(6 * p[..., 1] + 1), np.sin(6 * p[..., 2] + 2)], -1)
  shade = np.clip((p * np.array([0.5, 0.3, 0.8])).sum(-1, keepdims=True) * 0.5 + 0.6, 0.2, 1.0)
  rgb = np.where(hit[..., None], tex * shade, 1.0).astype(np.float32)
  f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(device)
  shp = (n_patch, P, P)
  rays = utils.Rays(pix_coords=f32(np.zeros(shp + (2,))), origins=f32(o), directions=f32(d), viewdirs=f32(v),
                    radii=f32(np.full(shp + (1,), 0.004 * 2 / np.sqrt(12))), lossmult=f32(np.ones(shp + (1,))),
                    static_mask=f32(np.ones(shp + (1,))), near=f32(np.full(shp + (1,), 1.5)), far=f32(np.full(shp + (1,), 4.5)),
                    embed_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device),
                    cam_idx=torch.zeros(shp + (1,), dtype=torch.int32, device=device))
  return utils.Batch(rays=rays, rgb=f32(rgb))


def run(dtype, steps):
  configs.clear_config(); configs.parse_config_files_and_bindings(None, GIN)
  config = configs.make_config()
  model, state, _, train_step, _ = train_utils.setup_model(config, 1234, compute_dtype=dtype)
  rng = np.random.default_rng(0)
  gen = torch.Generator(device='cuda').manual_seed(0)
  out = []
  val = scene_batch(np.random.default_rng(999), 16, 16, 'cuda')      # fixed 4096-ray validation set
  t0 = time.time()
  for s in range(steps):
    batch = scene_batch(rng, 4, 16, 'cuda')
    state, stats, gen = train_step(gen, state, batch, s / (config.max_steps - 1), None)
    if s % 50 == 0 or s == steps - 1:
      rend, _ = model.apply(state.flat, None, val.rays, s / (config.max_steps - 1), False, refresh_weights=False)
      mse = float(((rend[-1]['rgb'] - val.rgb)**2).mean())
      out.append((s, -10 * np.log10(mse), float(stats['loss'])))
  torch.cuda.synchronize()
  return out, time.time() - t0


if __name__ == '__main__':
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
  a, ta = run('bf16', steps); b, tb = run('fp32', steps)
  print(f'# train PSNR trajectory, 1024 rays/step, NerfMLP 8x512 + PropMLP 4x256, {steps} steps; bf16 {ta:.1f}s, fp32 {tb:.1f}s')
  print('# step   valPSNR_bf16  valPSNR_fp32   diff_dB    trainloss_bf16   trainloss_fp32   (PSNR on a fixed 4096-ray validation set)')
  for (s, pa, la), (_, pb, lb) in zip(a, b):
    print(f'{s:6d}  {pa:9.3f}  {pb:9.3f}  {pa-pb:+8.3f}  {la:10.6f}  {lb:10.6f}')
