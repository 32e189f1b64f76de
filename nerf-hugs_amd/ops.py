"""PyTorch custom-op registration of the hot-path kernels (`torch.ops.hugs.*`, torch.library): the form BASELINE.json's
north_star names -- "hand-written CDNA4 HIP invoked from PyTorch-ROCm custom ops through a thin C-ABI".  Each op is a
thin wrapper over the same `extern "C"` entry point of libhugs_hip.so that `_lib.call` binds (include/hugs.h); they
mutate caller-provided output tensors (no allocation inside), are CUDA-only (there is no CPU fallback) and carry no
autograd formula: the backward pass of this build is hand-scheduled (internal/engine.py), as the reference's is
whatever jax.value_and_grad derives.  `internal/engine.py` itself calls `_lib.call` directly (1.6 ms of Python per
step instead of ~3 with the dispatcher in between); the ops are the surface for callers that want to compose the
kernels from PyTorch code.

    import nerf_hugs_amd.ops                      # registers the namespace
    torch.ops.hugs.gemm_nt(x, wt, bias, True, out)
"""
from typing import Optional

import torch
from torch import Tensor

from . import _lib
from .internal import stepfun as _stepfun

_CU = 'cuda'


def _dt(t):
  if t.dtype == torch.bfloat16:
    return 1
  if t.dtype == torch.float32:
    return 0
  raise TypeError(f'hugs ops take float32 or bfloat16 matrices, got {t.dtype}')


@torch.library.custom_op('hugs::level_sample', mutates_args=('sdist', 'tdist'), device_types=_CU)
def level_sample(t_prev: Tensor, w_prev: Tensor, do_dilate: bool, dilation: float, anneal: float, resample_padding: float,
                 u_base: Tensor, jitter: Optional[Tensor], raydist: int, near: Tensor, far: Tensor, sdist: Tensor,
                 tdist: Tensor) -> None:
  """models.py:155-212 sampling level (hugs_level_sample_fwd)."""
  N, ns = sdist.shape[0], sdist.shape[1] - 1
  _lib.call('hugs_level_sample_fwd', N, t_prev, w_prev, w_prev.shape[1], int(do_dilate), dilation, 0., 1., anneal,
            resample_padding, u_base, jitter, 1, ns, raydist, _stepfun.SUM_ORDER, near, far, sdist, tdist, None, None, None)


@torch.library.custom_op('hugs::cast_ipe', mutates_args=('out',), device_types=_CU)
def cast_ipe(tdist: Tensor, origins: Tensor, directions: Tensor, radii: Tensor, basis: Tensor, ray_shape: int, warp: bool,
             max_deg: int, out: Tensor) -> None:
  """render.py:103-127 + coord.py:21-60,102-133: rays -> Gaussians -> (contract) -> lift -> IPE rows (hugs_cast_ipe_fwd)."""
  N, S = tdist.shape[0], tdist.shape[1] - 1
  _lib.call('hugs_cast_ipe_fwd', N, S, tdist, origins, directions, radii, basis, basis.shape[1], ray_shape, int(warp), max_deg,
            _dt(out), out.shape[1], out)


@torch.library.custom_op('hugs::gemm_nt', mutates_args=('out',), device_types=_CU)
def gemm_nt(x: Tensor, wt: Tensor, bias: Optional[Tensor], relu: bool, out: Tensor) -> None:
  """out[M,N] = act(x[M,K] @ wt[N,K]^T + bias): a Dense layer, models.py:451-456 (hugs_gemm_nt)."""
  M, K = x.shape
  _lib.call('hugs_gemm_nt', _dt(x), M, wt.shape[0], K, 0, x, K, None, 0, wt, K, bias, None, 1, 0, int(relu), None, 0, None, None,
            out, out.shape[1])


@torch.library.custom_op('hugs::gemm_nt_masked', mutates_args=('out',), device_types=_CU)
def gemm_nt_masked(g: Tensor, wn: Tensor, y: Tensor, out: Tensor) -> None:
  """out[M,K] = (g[M,N] @ wn[K,N]^T) * (y > 0): the input gradient through a relu layer (hugs_gemm_nt with a mask)."""
  M, N = g.shape
  _lib.call('hugs_gemm_nt', _dt(g), M, wn.shape[0], N, 0, g, N, None, 0, wn, N, None, None, 1, 0, 0, y, y.shape[1], None, None,
            out, out.shape[1])


@torch.library.custom_op('hugs::gemm_tn', mutates_args=('dw', 'db', 'workspace'), device_types=_CU)
def gemm_tn(x: Tensor, g: Tensor, nsplit: int, dw: Tensor, db: Optional[Tensor], workspace: Tensor) -> None:
  """dw[K,N] = x[M,K]^T @ g[M,N], db = colsum(g): the weight gradient of a Dense layer (hugs_gemm_tn)."""
  M, K = x.shape
  need = _lib.lib().cdll.hugs_gemm_tn_ws_bytes(K, g.shape[1], nsplit)
  if workspace.numel() * workspace.element_size() < need:
    raise ValueError(f'hugs::gemm_tn: workspace of {need} bytes needed')
  _lib.call('hugs_gemm_tn', _dt(x), M, K, g.shape[1], nsplit, x, K, g, g.shape[1], dw, db, workspace)


@torch.library.custom_op('hugs::composite', mutates_args=('weights', 'rgb_out'), device_types=_CU)
def composite(density: Tensor, rgb_s: Optional[Tensor], tdist: Tensor, directions: Tensor, opaque_background: bool, bg: float,
              weights: Tensor, rgb_out: Tensor) -> None:
  """render.py:130-151,185-244 alpha compositing (hugs_composite_fwd)."""
  N, S = weights.shape
  _lib.call('hugs_composite_fwd', N, S, density, rgb_s, tdist, directions, int(opaque_background), bg, None, weights, rgb_out, None)


@torch.library.custom_op('hugs::composite_backward', mutates_args=('d_density', 'd_rgb_s'), device_types=_CU)
def composite_backward(density: Tensor, rgb_s: Optional[Tensor], tdist: Tensor, directions: Tensor, opaque_background: bool,
                       bg: float, d_rgb_out: Optional[Tensor], d_w_extra: Optional[Tensor], d_density: Tensor,
                       d_rgb_s: Optional[Tensor]) -> None:
  """Backward of hugs::composite (hugs_composite_bwd)."""
  N, S = tdist.shape[0], tdist.shape[1] - 1
  _lib.call('hugs_composite_bwd', N, S, density, rgb_s, tdist, directions, int(opaque_background), bg, d_rgb_out, d_w_extra,
            d_density, d_rgb_s)


OPS = ('level_sample', 'cast_ipe', 'gemm_nt', 'gemm_nt_masked', 'gemm_tn', 'composite', 'composite_backward')
