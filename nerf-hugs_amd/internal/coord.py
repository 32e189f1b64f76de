"""reference MipNeRF360/internal/coord.py entry points that exist as stand-alone kernels."""
import torch

from .. import _lib


def pos_enc(x, min_deg, max_deg, append_identity=True):
  """coord.py:136-147 for 3-vectors (the view-direction encoding)."""
  if min_deg != 0 or not append_identity or x.shape[-1] != 3:
    raise NotImplementedError('pos_enc is built for min_deg=0, append_identity=True, 3-vectors (viewdirs)')
  v = x.reshape(-1, 3).to(torch.float32).contiguous()
  out = torch.empty(v.shape[0], 3 + 6 * max_deg, device=v.device)
  _lib.call('hugs_dir_enc_fwd', v.shape[0], max_deg, v, out)
  return out.reshape(x.shape[:-1] + (3 + 6 * max_deg,))
