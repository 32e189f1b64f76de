"""GPU parity of the eval metrics (hugs_ssim, hugs_mse via image.MetricHarness) against oracle/image_ref.py, and the
eval.py output set written by nerf_hugs_amd.eval.evaluate."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(11, 11, 1), (19, 23, 3), (64, 50, 3), (240, 317, 3), (11, 200, 4)])
def test_ssim_and_psnr_vs_oracle(shape):
  from nerf_hugs_amd.internal import image
  from oracle import image_ref as I
  rng = np.random.default_rng(sum(shape))
  a = rng.uniform(size=shape).astype(np.float32)
  for noise in (0.0, 0.02, 0.3):
    b = np.clip(a + rng.normal(size=shape).astype(np.float32) * noise, 0, 1).astype(np.float32)
    if noise == 0:
      b = a.copy(); b[0, 0, 0] = 1 - b[0, 0, 0]
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert abs(float(image.ssim(ta, tb)) - I.ssim(a, b)) < 2e-5            # fp32 windows vs the float64 oracle
    m = image.MetricHarness()(ta, tb)
    assert abs(m['psnr'] - I.psnr(a, b)) < 1e-3 and abs(m['ssim'] - I.ssim(a, b)) < 2e-5
  assert abs(float(image.ssim(ta, ta)) - 1) < 1e-6
  flat = torch.full(shape, .25, device='cuda')                              # zero variance: the eps^2 clamp path
  assert abs(float(image.ssim(flat, flat)) - 1) < 1e-6


def test_ssim_errors_and_full_size_properties():
  from nerf_hugs_amd.internal import image
  with pytest.raises(ValueError):
    image.ssim(torch.zeros(10, 40, 3, device='cuda'), torch.zeros(10, 40, 3, device='cuda'))
  with pytest.raises(ValueError):
    image.ssim(torch.zeros(20, 40, 3, device='cuda'), torch.zeros(20, 41, 3, device='cuda'))
  g = torch.Generator(device='cuda').manual_seed(0)
  a = torch.rand(1080, 1920, 3, device='cuda', generator=g)
  b = (a + 0.05 * torch.randn(a.shape, device='cuda', generator=g)).clamp(0, 1)
  s_ab, s_ba = float(image.ssim(a, b)), float(image.ssim(b, a))
  assert s_ab == s_ba and 0 < s_ab < 1                                       # symmetric, bit for bit
  assert float(image.ssim(a, b)) == s_ab                                     # deterministic reduction order
  assert float(image.ssim(a, (a + b) / 2)) > s_ab                            # closer image scores higher
  assert abs(float(image.mse(a, b)) - float(((a - b).double() ** 2).mean())) < 1e-8


def test_evaluate_writes_the_reference_output_set(tmp_path):
  from PIL import Image
  from nerf_hugs_amd import eval as hugs_eval
  from nerf_hugs_amd.internal import configs, datasets, train_utils
  from oracle import image_ref as I
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, [
      "Config.patch_size = 1", "Config.batch_size = 64", "Config.image_num_per_batch = 1", "Model.num_levels = 3",
      "PropMLP.net_depth = 2", "PropMLP.net_width = 128", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 4",
      "NerfMLP.net_width = 128", "NerfMLP.bottleneck_width = 128", "Config.near = 0.5", "Config.far = 6.",
      "Config.render_chunk_size = 256", "Config.eval_crop_borders = 1", f"Config.checkpoint_dir = '{tmp_path}'"])
  config = configs.make_config()
  rng = np.random.default_rng(0)
  h, w, n = 20, 24, 3
  k = np.array([[30., 0, w / 2], [0, 30., h / 2], [0, 0, 1.]])
  c2w = np.stack([np.concatenate([np.eye(3), [[0.1 * i], [0.], [2.]]], 1) for i in range(n)])
  ds = datasets.ArrayDataset(config, images=[rng.uniform(size=(h, w, 3)).astype(np.float32) for _ in range(n)],
                             pixtocams=np.linalg.inv(k), camtoworlds=c2w, is_training=False, split='test')
  model, state, render_fn, _, _ = train_utils.setup_model(config, 0, compute_dtype='fp32')
  metrics, metrics_cc, times = hugs_eval.evaluate(config, ds, render_fn, state, step=7)
  out = os.path.join(str(tmp_path), 'test_preds')
  names = set(os.listdir(out))
  for i in range(n):
    for suf in ('color.png', 'gt.png', 'colorcc.png', 'acc.tiff'):
      assert f'{i:04d}_{suf}' in names
  for f in ('metric_psnr_7.txt', 'metric_ssim_7.txt', 'metric_cc_psnr_7.txt', 'metric_cc_ssim_7.txt', 'metric_mean_7.txt',
            'render_times_7.txt'):
    assert f in names
  assert len(open(os.path.join(out, 'metric_psnr_7.txt')).read().split()) == n
  # the PNGs on disk reproduce the metrics up to save_img_u8's truncation (utils.py:157-162 floors, the metrics round:
  # the reference has the same half-level offset)
  pred = np.asarray(Image.open(os.path.join(out, '0001_color.png')), np.float64) / 255
  gt = np.asarray(Image.open(os.path.join(out, '0001_gt.png')), np.float64) / 255
  assert pred.shape == (h, w, 3)
  gt_full = np.asarray(ds.images[1], np.float64)
  assert np.abs(gt - gt_full).max() <= 1 / 255
  assert abs(I.psnr(pred[1:-1, 1:-1], gt_full[1:-1, 1:-1]) - metrics[1]['psnr']) < 0.05
  assert abs(I.ssim(pred[1:-1, 1:-1], gt_full[1:-1, 1:-1]) - metrics[1]['ssim']) < 0.01
  assert metrics_cc[1]['psnr'] >= metrics[1]['psnr'] - 1e-6        # colour correction never hurts the fit
  mean = open(os.path.join(out, 'metric_mean_7.txt')).read()
  assert 'psnr:' in mean and 'cc_ssim:' in mean and 'render time:' in mean
