"""Batch-assembly throughput on the device: ArrayDataset.__next__ (train patches) and generate_ray_batch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from nerf_hugs_amd.internal import configs, datasets

def scene(n, h, w):
  rng = np.random.default_rng(0)
  imgs = [rng.integers(0, 256, (h, w, 3)).astype(np.uint8) for _ in range(n)]
  masks = [(rng.uniform(size=(h, w, 1)) < .8).astype(np.float32) for _ in range(n)]
  k = np.array([[1.2 * w, 0, w / 2], [0, 1.2 * w, h / 2], [0, 0, 1.]])
  c2w = np.stack([np.concatenate([np.linalg.qr(rng.normal(size=(3, 3)))[0], rng.normal(size=(3, 1))], 1) for _ in range(n)])
  return dict(images=imgs, static_masks=masks, pixtocams=np.linalg.inv(k).astype(np.float32), camtoworlds=c2w.astype(np.float32))

for bs, ps, ipb in [(1024, 16, 4), (4096, 16, 16), (8192, 16, 32), (65536, 1, 64)]:
  configs.clear_config()
  config = configs.make_config(batch_size=bs, patch_size=ps, image_num_per_batch=ipb)
  ds = datasets.ArrayDataset(config, random_state=np.random.RandomState(0), distortion_params=dict(k1=-.05, k2=.01), **scene(64, 768, 1024))
  for _ in range(20): next(ds)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(200): b = next(ds)
  t_host = time.perf_counter() - t0
  torch.cuda.synchronize(); t = time.perf_counter() - t0
  print(f'train batch {bs:6d} rays (patch {ps}, {ipb} images): {t / 200 * 1e3:.3f} ms/batch ({t_host / 200 * 1e3:.3f} ms host enqueue) = {bs * 200 / t / 1e6:.2f} M rays/s')
ds.is_training = False
for _ in range(3): ds.generate_ray_batch(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): b = ds.generate_ray_batch(i)
torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f'full image 768x1024 (undistort on): {t / 20 * 1e3:.3f} ms = {768 * 1024 * 20 / t / 1e6:.1f} M rays/s')
