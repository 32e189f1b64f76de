"""hugs_gemm_nt_chain (csrc/hugs_gemm_chain.inc; HUGS_NT_CHAIN=1, off by default -- measured slower than the per-layer launches):
all trunk layers of the NerfMLP as ONE persistent launch whose workgroups hand row bands from layer to layer through per-band counters.
Every trunk activation and every 1-bit relu mask word must equal the per-layer launches' bit for bit; a shape that does not qualify is
refused with the library's error, not launched."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', ['cfg2', 'ref360'])
def test_chained_trunk_is_bit_identical_to_the_per_layer_launches(shape):
  """cfg2: 1024 rays x 128 samples (8 tiles per CU and layer); ref360: the reference-default shape, 16 384 rays x 32 samples at the third
  level (32 tiles per CU and layer), contraction + reciprocal ray distance."""
  import bench
  from nerf_hugs_amd.internal import configs, train_utils, engine as E, models as M
  from tests import hugs_testlib as H
  configs.clear_config(); configs.parse_config_files_and_bindings(None, bench.GIN if shape == 'cfg2' else bench.GIN_REF360)
  config = configs.make_config()
  model, state, _, _, _ = train_utils.setup_model(config, 0, compute_dtype='bf16')
  n_rays = 1024 if shape == 'cfg2' else 16384
  batch = H.synth_rays(n_rays // 256, 16, 7) if shape == 'cfg2' else H.synth_rays(n_rays // 256, 16, 7, near=0.2, far=1e6)
  eng = model.engine('cuda')
  rays = M.rays_to_dict(batch.rays, 'cuda')
  gen = torch.Generator(device='cuda').manual_seed(1)
  u01 = [torch.rand(n_rays, generator=gen, device='cuda') for _ in range(model.num_levels)]
  tag = f'NerfMLP_0/L{model.num_levels - 1}/'
  pick = lambda: {k[0]: t for k, t in eng.ws.bufs.items()
                  if torch.is_tensor(t) and isinstance(k[0], str) and k[0].startswith(tag) and ('/Y' in k[0] or 'bits' in k[0])}
  out = {}
  old = E._NT_CHAIN
  try:
    for mode in (False, True):
      E._NT_CHAIN = mode
      eng.refresh_weights(state.flat)
      for t in pick().values():
        t.zero_()
      levels = eng.forward(state.flat, rays, 0.5, u01, False)
      torch.cuda.synchronize()
      out[mode] = {k: t.clone() for k, t in pick().items()}
      out[mode]['rgb'] = levels[-1]['rgb_out'].clone()
    assert len(out[True]) == 17                                   # 8 activations + 8 mask-bit buffers + the rendered colour
    assert any(('nt_chain',) == k[:1] for k in eng.ws.bufs)        # the chained launch did run
    for k in out[False]:
      assert int((out[True][k] != 0).sum()) > 0, k
      assert torch.equal(out[False][k], out[True][k]), k
  finally:
    E._NT_CHAIN = old


def test_chain_refuses_shapes_it_cannot_walk():
  from nerf_hugs_amd import _lib
  tab = np.zeros((2, 12), np.uint64)
  flags = torch.zeros(64, dtype=torch.int32, device='cuda')
  with pytest.raises(_lib.HugsError):
    _lib.call('hugs_gemm_nt_chain', 1, 4096, 1024, 2, tab.ctypes.data, flags)      # 64 tiles: not a whole number >= 4 per CU
  with pytest.raises(_lib.HugsError):
    _lib.call('hugs_gemm_nt_chain', 0, 131072, 1024, 2, tab.ctypes.data, flags)    # fp32
