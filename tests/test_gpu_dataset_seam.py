"""The f1 / f2 seam end to end on the GPU: an on-disk mini scene (tests/golden/loader_scenes/kubric_mini, the files the
reference's own Kubric loader was run on for ref_loaders.npz) -> loaders.load_dataset(device='cuda') (images, masks and
camera tables resident in HBM) -> Dataset.__next__ (device-side patch sampling + ray generation) -> train_step, and the
test split -> eval.evaluate (render_image + metrics + the experiment-directory files)."""
import os

import numpy as np
import pytest
import torch

import json

from PIL import Image

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, 'golden', 'loader_scenes', 'kubric_mini')
GIN = ["Config.dataset_loader = 'kubric'", "Config.batch_size = 128", "Config.patch_size = 8", "Config.image_num_per_batch = 2",
       "Config.data_loss_type = 'mse'", "Config.max_steps = 100", "Config.render_chunk_size = 64", "Config.eval_save_output = True",
       "Model.opaque_background = True", "Model.num_levels = 2", "Model.num_prop_samples = 64", "Model.num_nerf_samples = 32",
       "PropMLP.net_depth = 2", "PropMLP.net_width = 128", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 2",
       "NerfMLP.net_width = 128", "Model.num_glo_features = 4"]


def _write_kubric_scene(root, H, W, n_train, n_test, seed):
  """A Kubric-format scene directory (datasets.py:984-1113 layout: rgb/1x/*.png, camera-gt/*.json, static_masks/*.png,
  scene_gt.json, dataset.json, freeze-test/...) with images large enough for the 11-tap SSIM window."""
  rng = np.random.default_rng(seed)
  os.makedirs(os.path.join(root, 'freeze-test'), exist_ok=True)
  json.dump(dict(center=[0., 0., 0.], scale=0.5, near=0.4, far=2.5), open(os.path.join(root, 'scene_gt.json'), 'w'))
  json.dump(dict(train_ids=list(range(n_train))), open(os.path.join(root, 'dataset.json'), 'w'))
  json.dump(dict(val_ids=list(range(n_test))), open(os.path.join(root, 'freeze-test', 'dataset.json'), 'w'))
  yy, xx = np.mgrid[0:H, 0:W]
  for sub, n, rgbdir in ((root, n_train, 'rgb/1x'), (os.path.join(root, 'freeze-test'), n_test, 'static-rgb/1x')):
    for i in range(n):
      img = np.stack([127 + 100 * np.sin(0.3 * xx + i), 127 + 100 * np.cos(0.2 * yy - i), 60 + 5 * xx], -1).clip(0, 255).astype(np.uint8)
      for d, arr in ((rgbdir, img), ('static_masks', np.stack([(rng.uniform(size=(H, W)) < 0.7).astype(np.uint8) * 255] * 3, -1))):
        os.makedirs(os.path.join(sub, d), exist_ok=True)
        Image.fromarray(arr).save(os.path.join(sub, d, f'{i}.png'), 'PNG')
      th = 2 * np.pi * i / max(n, 1)
      c, s_ = np.cos(th), np.sin(th)
      os.makedirs(os.path.join(sub, 'camera-gt'), exist_ok=True)
      json.dump(dict(orientation=[[c, -s_, 0.], [s_, c, 0.], [0., 0., 1.]], position=[2 * c, 2 * s_, 0.3], focal_length=float(W),
                     principal_point=[W / 2, H / 2], skew=0., pixel_aspect_ratio=1., radial_distortion=[0., 0., 0.],
                     tangential_distortion=[0., 0.]), open(os.path.join(sub, 'camera-gt', f'{i}.json'), 'w'))


def test_disk_scene_to_train_step_and_evaluate(tmp_path):
  from nerf_hugs_amd import eval as hugs_eval
  from nerf_hugs_amd.internal import configs, loaders, train_utils
  from nerf_hugs_amd.internal import random as hrandom
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, GIN)
  config = configs.make_config(checkpoint_dir=str(tmp_path), data_dir=SCENE)
  gold = np.load(os.path.join(HERE, 'golden', 'ref_loaders.npz'))
  train = loaders.load_dataset('train', True, False, config.batch_size, config.patch_size, config.patch_dilation,
                               config.image_num_per_batch, SCENE, config, device='cuda', random_state=np.random.RandomState(7))
  # the test split comes from a scene written here in the same on-disk format: the committed mini scenes hold 10 x 12
  # images, smaller than the SSIM window eval.py's metrics need (image.py / dm_pix: 11 taps)
  big = str(tmp_path / 'kubric_24x32')
  _write_kubric_scene(big, 24, 32, 2, 2, 1)
  test = loaders.load_dataset('test', False, False, config.batch_size, config.patch_size, config.patch_dilation,
                              config.image_num_per_batch, big, config, device='cuda')
  assert train._images.is_cuda and test._images.is_cuda and train.size == 3 and test.size == 2
  # the HBM-resident pixels are the reference loader's own decode of these files
  h, w = int(train.heights[0]), int(train.widths[0])
  img0 = train._images[:h * w].float().cpu().numpy().reshape(h, w, 3)
  img0 = img0 / 255.0 if train._images.dtype == torch.uint8 else img0
  np.testing.assert_allclose(img0, gold['kubric_train/images/0'], atol=1e-6)
  model, state, render_eval_pfn, train_pstep, _ = train_utils.setup_model(config, 0, compute_dtype='fp32')
  key = hrandom.PRNGKey(3, 'cuda')
  theta0 = state.flat.clone()
  losses = []
  for step in range(3):
    batch = next(train)
    assert batch.rays.origins.is_cuda and batch.rays.origins.shape == (2, 8, 8, 3) and batch.rgb.shape == (2, 8, 8, 3)
    assert float(batch.rgb.min()) >= 0 and float(batch.rgb.max()) <= 1
    # every ray of a patch comes from one camera and carries that image's embedding index (datasets.py:494-548)
    ci = batch.rays.cam_idx.reshape(2, -1)
    assert bool((ci == ci[:, :1]).all())
    state, stats, key = train_pstep(key, state, batch, step / config.max_steps, None)
    losses.append(float(stats['loss']))
  assert np.isfinite(losses).all() and bool((state.flat != theta0).any()) and state.step == 3
  metrics, metrics_cc, times = hugs_eval.evaluate(config, test, render_eval_pfn, state, out_dir=str(tmp_path / 'test_preds'))
  assert len(metrics) == 2 and all(np.isfinite(m['psnr']) and 0 < m['psnr'] < 60 and -1 <= m['ssim'] <= 1 for m in metrics)
  for name in ('0000_color.png', '0000_gt.png', '0000_colorcc.png', '0000_acc.tiff', 'metric_psnr_3.txt', 'metric_ssim_3.txt',
               'metric_cc_psnr_3.txt', 'metric_mean_3.txt', 'render_times_3.txt'):
    assert os.path.exists(tmp_path / 'test_preds' / name), name
  # the ground-truth image eval wrote is the test image on disk
  gt = loaders.load_img(str(tmp_path / 'test_preds' / '0000_gt.png'))[..., :3]
  src = loaders.load_img(os.path.join(big, 'freeze-test', 'static-rgb', '1x', '0.png'))[..., :3]
  assert gt.shape == (24, 32, 3) and float(np.abs(gt - src).max()) <= 1.0      # save_img_u8 truncates to uint8
  configs.clear_config()
