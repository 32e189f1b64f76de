"""k_gemm_nt_bf16_p64 (csrc/hugs_gemm_p64.inc, round 6): the persistent 256x256 NT kernel with K staged in 64-wide super-stages of whole
cache lines.  It must reproduce k_gemm_nt_bf16_pers (HUGS_NT_K64=0: 32-wide stages) BIT FOR BIT -- same K rotation per tile, same
accumulation order -- for every epilogue specialisation the dispatcher sends to it, on shapes with 2, 4 and 6 super-stage pairs, a
two-segment K (the skip layer), tile counts that are not a multiple of the CU count, and against float64."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = 'cuda'


def _L():
  from nerf_hugs_amd import _lib
  return _lib


def _both(fn):
  """fn() under HUGS_NT_K64=0, =1 and =1 with HUGS_NT_W4=1 (the four-wave form: present only in a library built with -DHUGS_BUILD_W4 --
  measured slower, profiles/r06_nt_w4_ab.txt; in the shipped library the third run repeats the second; the library reads the switches
  per call); returns the three result lists."""
  old = {k: os.environ.get(k) for k in ('HUGS_NT_K64', 'HUGS_NT_W4')}
  out = []
  try:
    for v, w in (('0', '0'), ('1', '0'), ('1', '1')):
      os.environ['HUGS_NT_K64'], os.environ['HUGS_NT_W4'] = v, w
      out.append(fn())
      torch.cuda.synchronize()
  finally:
    for k, v in old.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v
  return out


@pytest.mark.parametrize('shape', [(131072, 1024, 1024, 0), (66560, 1024, 512, 0), (66560, 256, 512, 0), (67072, 512, 1024, 512),
                                   (66560, 256, 1024, 0), (133120, 256, 512, 0)])
def test_k64_kernel_bit_identical_to_k32_all_epilogues(shape):
  L = _L()
  M, N, K1, K2 = shape
  K = K1 + K2
  g = torch.Generator(device=dev).manual_seed(M + N + K)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  A1 = rn(M, K1).clamp(min=0).bfloat16()                      # post-relu activations, as the trunk's
  A2 = rn(M, K2).bfloat16() if K2 else None
  Bt = (rn(N, K) / K**0.5).bfloat16()
  bias, r1r, r1c = rn(N), rn(M), rn(N)
  mk = rn(M, N).bfloat16()

  def run():
    res = []
    y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); bits = torch.zeros(M * N // 32, dtype=torch.int32, device=dev)
    L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, bias, 1, None, None, y, N, bits, None)          # EPI 35
    res += [y, bits]
    for r1 in (False, True):                                                                                              # EPI 16 / 24
      o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
      L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, None, 0, r1r if r1 else None, r1c if r1 else None, o, N, None, bits)
      res.append(o)
    for (b_, relu_, mask_, r1_) in ((True, 1, None, False), (True, 0, None, False), (False, 0, None, False), (False, 0, mk, False),
                                    (False, 0, mk, True)):                                                                # EPI 3, 1, 0, 4, 12
      o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
      L.call('hugs_gemm_nt', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, bias if b_ else None, None, 1, 0, relu_, mask_, N,
             r1r if r1_ else None, r1c if r1_ else None, o, N)
      res.append(o)
    return res

  a, b, c = _both(run)
  for i, (x, y, z) in enumerate(zip(a, b, c)):
    assert int((y != 0).sum()) > 0, i
    assert torch.equal(x, y), f'output {i} differs between the 32-wide and the 64-wide staging'
    assert torch.equal(x, z), f'output {i} differs between the eight-wave and the four-wave kernel'
  A = torch.cat([A1, A2], 1) if K2 else A1
  rows = slice(0, 4096)
  ref = (A[rows].double() @ Bt.double().T + bias.double()).clamp(min=0)
  assert float((b[0][rows].double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
  ref = (A[rows].double() @ Bt.double().T + r1r[rows].double()[:, None] * r1c.double()[None]) * (b[0][rows].double() > 0)
  assert float((b[3][rows].double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))


def test_k64_kernel_is_the_one_that_ran():
  """The cycle account keys on (specialisation, K class) for both kernels; the 64-wide kernel is selected by default for the trunk
  shape and the account sees its tiles."""
  L = _L()
  M, N, K = 66560, 1024, 1024
  A = torch.randn(M, K, device=dev).bfloat16(); Bt = torch.randn(N, K, device=dev).bfloat16(); bias = torch.randn(N, device=dev)
  y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bits = torch.empty(M * N // 32, dtype=torch.int32, device=dev)
  cyc = torch.zeros(64 * 4 * 2, dtype=torch.int64, device=dev)
  torch.cuda.synchronize()
  L.call('hugs_debug_set_nt_cycles', cyc.data_ptr())
  try:
    L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, 1, None, None, y, N, bits, None)
    torch.cuda.synchronize()
  finally:
    L.call('hugs_debug_set_nt_cycles', 0)
  c = cyc.cpu().reshape(64, 4, 2)
  assert int(c[35, 2, 1]) == (M // 256) * (N // 256) and int(c[35, 2, 0]) > 0
  assert int(c.sum()) == int(c[35, 2].sum())


@pytest.mark.parametrize('M', [16384, 8192, 4096 + 256])
def test_k64_kernel_on_grids_of_at_most_one_tile_per_cu(M):
  """Round 6: grids of at most one 256 x 256 tile per CU (the 128-ray step's trunk layers: 256 tiles) run the whole-line kernel with one
  tile per workgroup instead of k_gemm_nt_bf16_big (HUGS_NT_P64_SMALL=0).  The two differ in fp32 summation order (K rotation per row
  band) and in nothing else: forward with mask bits and the dX form against each other and against float64."""
  L = _L()
  N, K = 1024, 1024
  g = torch.Generator(device=dev).manual_seed(M)
  A = torch.randn(M, K, device=dev, generator=g).clamp(min=0).bfloat16()
  G = torch.randn(M, K, device=dev, generator=g).bfloat16()
  Bt = (torch.randn(N, K, device=dev, generator=g) / K**0.5).bfloat16()
  bias = torch.randn(N, device=dev, generator=g)
  out = {}
  old = os.environ.get('HUGS_NT_P64_SMALL')
  try:
    for mode in ('0', '1'):
      os.environ['HUGS_NT_P64_SMALL'] = mode
      y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); bits = torch.zeros(M * N // 32, dtype=torch.int32, device=dev)
      o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
      L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, 1, None, None, y, N, bits, None)
      L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, G, K, None, 0, Bt, K, None, 0, None, None, o, N, None, bits)
      torch.cuda.synchronize()
      out[mode] = (y, bits, o)
  finally:
    if old is None:
      os.environ.pop('HUGS_NT_P64_SMALL', None)
    else:
      os.environ['HUGS_NT_P64_SMALL'] = old
  ref = (A.double() @ Bt.double().T + bias.double()).clamp(min=0)
  for mode in ('0', '1'):
    y, bits, o = out[mode]
    assert float((y.double() - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
    refx = (G.double() @ Bt.double().T) * (y.double() > 0)
    assert float((o.double() - refx).abs().max()) < 2e-2 * max(1.0, float(refx.abs().max()))
  # one bf16 rounding apart at most, on a handful of entries (a different fp32 summation order in front of the rounding)
  d = (out['0'][0].float() - out['1'][0].float()).abs()
  assert float(d.max()) <= 2.0 ** -7 * float(out['0'][0].float().abs().max()) and float((d > 0).float().mean()) < 0.05
