"""The per-checkpoint body of the reference's eval.py (MipNeRF360/eval.py:83-228): render every test image, score
it (PSNR / SSIM, raw and colour-corrected), and write the output set a HuGS experiment directory holds --
`NNNN_color.png`, `NNNN_gt.png`, `NNNN_colorcc.png`, `NNNN_acc.tiff`, `NNNN_distance_{mean,median}.tiff`,
`metric_<name>_<step>.txt`, `metric_cc_<name>_<step>.txt`, `metric_mean_<step>.txt`, `render_times_<step>.txt` --
which is what the repository-level `metrics.py` / `scripts/metrics_*.sh` then read.

Not rebuilt: flag parsing, the poll-for-new-checkpoints loop, tensorboard summaries and the turbo-colormapped depth
PNGs (matplotlib): control plane / visualisation, out of scope.  Rendering and metrics run on the GPU.
"""
import functools
import os
import time

import numpy as np
import torch

from .internal import image
from .internal import models
from .internal import utils


def evaluate(config, dataset, render_eval_fn, state, step=None, out_dir=None, verbose=False):
  """eval.py:83-228 for one checkpoint.  `render_eval_fn` and `state` come from train_utils.setup_model;
  `dataset` is a test-split internal.datasets.Dataset.  Returns (metrics, metrics_cc, render_times)."""
  step = int(state.step) if step is None else int(step)
  if out_dir is None:
    out_dir = os.path.join(config.checkpoint_dir, 'test_preds')
  path_fn = lambda x: os.path.join(out_dir, x)
  rank0 = (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0
  if config.eval_save_output and rank0:
    os.makedirs(out_dir, exist_ok=True)
  harness = image.MetricHarness(state.flat.device)
  num_eval = min(dataset.size, config.eval_dataset_limit)
  metrics, metrics_cc, render_times = [], [], []
  train_frac = float(np.clip(step / config.max_steps, 0, 1))
  for idx in range(dataset.size):
    t0 = time.time()
    batch = next(dataset)
    if idx >= num_eval:
      continue
    rendering = models.render_image(functools.partial(render_eval_fn, state.params, train_frac), batch.rays, None,
                                    config, verbose=verbose)
    if not rank0:
      continue
    torch.cuda.synchronize()
    render_times.append(time.time() - t0)
    gt_rgb = batch.rgb.detach().cpu().numpy().astype(np.float64)
    rgb = rendering['rgb'].detach().cpu().numpy().astype(np.float64)
    rgb_cc = image.color_correct(rgb, gt_rgb)
    m_rgb, m_cc, m_gt = rgb, rgb_cc, gt_rgb
    if config.eval_quantize_metrics:     # ensures that the images written to disk reproduce the metrics
      m_rgb = np.round(m_rgb * 255) / 255
      m_cc = np.round(m_cc * 255) / 255
    if config.eval_crop_borders > 0:
      c = config.eval_crop_borders
      m_rgb, m_cc, m_gt = m_rgb[c:-c, c:-c], m_cc[c:-c, c:-c], m_gt[c:-c, c:-c]
    metrics.append(harness(m_rgb, m_gt))
    metrics_cc.append(harness(m_cc, m_gt))
    if config.eval_save_output and config.eval_render_interval > 0 and idx % config.eval_render_interval == 0:
      utils.save_img_u8(rgb, path_fn(f'{idx:04d}_color.png'))
      utils.save_img_u8(gt_rgb, path_fn(f'{idx:04d}_gt.png'))
      utils.save_img_u8(rgb_cc, path_fn(f'{idx:04d}_colorcc.png'))
      for key in ('distance_mean', 'distance_median'):
        if key in rendering:
          utils.save_img_f32(rendering[key], path_fn(f'{idx:04d}_{key}.tiff'))
      utils.save_img_f32(rendering['acc'], path_fn(f'{idx:04d}_acc.tiff'))
  if config.eval_save_output and rank0 and metrics:
    with open(path_fn(f'render_times_{step}.txt'), 'w') as f:
      f.write(' '.join(str(r) for r in render_times))
    for name in metrics[0]:
      with open(path_fn(f'metric_{name}_{step}.txt'), 'w') as f:
        f.write(' '.join(str(m[name]) for m in metrics))
    for name in metrics_cc[0]:
      with open(path_fn(f'metric_cc_{name}_{step}.txt'), 'w') as f:
        f.write(' '.join(str(m[name]) for m in metrics_cc))
    with open(path_fn(f'metric_mean_{step}.txt'), 'w') as f:
      f.write(f'render time: {np.mean(render_times)}\n')
      for name in metrics[0]:
        f.write(f'{name}: {np.mean([m[name] for m in metrics])}\n')
      for name in metrics_cc[0]:
        f.write(f'cc_{name}: {np.mean([m[name] for m in metrics_cc])}\n')
  return metrics, metrics_cc, render_times
