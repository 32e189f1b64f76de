"""k_gemm_nt_bf16_dq (csrc/hugs_gemm_dq.inc, round 6): the persistent whole-line NT kernel drawing its tiles from per-XCD ticket counters
between hugs_gemm_nt_queue_begin / _end.  Which workgroup computes a tile must not matter: outputs and mask words bit-identical to the
static walk, with the chip to itself and with a kernel of another stream holding CUs when the launch starts; every tile computed exactly
once (the counters end at the band lengths); a region that runs out of slots falls back to the static walk."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = 'cuda'


def _run(M, N, K1, K2, queue, neighbour=False, nslots=8):
  from nerf_hugs_amd import _lib as L
  K = K1 + K2
  g = torch.Generator(device=dev).manual_seed(M + N + K)
  rn = lambda *s: torch.randn(*s, device=dev, generator=g)
  A1 = rn(M, K1).clamp(min=0).bfloat16(); A2 = rn(M, K2).bfloat16() if K2 else None
  G = rn(M, K).bfloat16()
  Bt = (rn(N, K) / K**0.5).bfloat16(); bias = rn(N); r1r, r1c = rn(M), rn(N)
  y = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); bits = torch.zeros(M * N // 32, dtype=torch.int32, device=dev)
  o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16); o2 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
  q = torch.full((nslots * 8,), 12345, dtype=torch.int32, device=dev)      # (begin zeroes it)
  side = torch.cuda.Stream()
  big = torch.empty(1 << 27, device=dev)
  if queue:
    L.call('hugs_gemm_nt_queue_begin', q, q.numel() * 4)
  try:
    if neighbour:
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        for _ in range(4):
          big.mul_(1.0001)
    L.call('hugs_gemm_nt_bits', 1, M, N, K1, K2, A1, K1, A2, K2, Bt, K, bias, 1, None, None, y, N, bits, None)
    L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, G, K, None, 0, Bt, K, None, 0, None, None, o, N, None, bits)
    L.call('hugs_gemm_nt_bits', 1, M, N, K, 0, G, K, None, 0, Bt, K, None, 0, r1r, r1c, o2, N, None, bits)
  finally:
    if queue:
      L.call('hugs_gemm_nt_queue_end', 0)
  torch.cuda.synchronize()
  return y, bits, o, o2, q


@pytest.mark.parametrize('shape', [(131072, 1024, 1024, 0), (133120, 1024, 1024, 512), (262144, 256, 512, 0)])
def test_tile_queue_is_bit_identical_to_the_static_walk(shape):
  M, N, K1, K2 = shape
  ref = _run(M, N, K1, K2, False)
  for neighbour in (False, True):
    got = _run(M, N, K1, K2, True, neighbour)
    for i in range(4):
      assert int((got[i] != 0).sum()) > 0
      assert torch.equal(ref[i], got[i]), (i, neighbour)
    q = got[4].cpu().view(-1, 8)
    ncu = torch.cuda.get_device_properties(0).multi_processor_count & ~7
    ntiles = (M // 256) * (N // 256)
    bq, br = ntiles >> 3, ntiles & 7
    for launch in range(3):      # three launches took three slots; every XCD's counter ends at its band length minus the static first tiles ... plus one miss per workgroup
      for x in range(8):
        assert int(q[launch, x]) == bq + (1 if x < br else 0), (launch, x, q[launch].tolist())
    assert int(q[3:].abs().sum()) == 0
  assert int(ref[4][0]) == 12345      # without begin / end nothing touches a queue


def test_tile_queue_out_of_slots_falls_back_to_the_static_walk():
  M, N, K = 131072, 1024, 1024
  ref = _run(M, N, K, 0, False)
  got = _run(M, N, K, 0, True, nslots=1)      # one slot for three launches
  for i in range(4):
    assert torch.equal(ref[i], got[i])
  assert int(got[4][:8].sum()) == (M // 256) * (N // 256)


def test_train_step_with_and_without_tile_queues_is_bit_identical():
  """The whole bf16 train step (full-width nets: trunk launches of 2048 tiles) with HUGS_NT_DYNQ on and off: same parameters after 3 steps."""
  from tests import hugs_testlib as H
  from tests.test_gpu_train_step import SMALL
  from nerf_hugs_amd.internal import train_utils
  gin = [g for g in SMALL if 'net_width' not in g and 'patch_size' not in g] + ["PropMLP.net_width = 256", "NerfMLP.net_width = 1024", "Config.patch_size = 16"]
  out = []
  old = train_utils._NT_DYNQ
  try:
    for on in (False, True):
      train_utils._NT_DYNQ = on
      config, model, state, _, train_step, cfg, oparams = H.make_pair(gin, compute_dtype='bf16')
      batch = H.synth_rays(4, 16, 5)
      for i in range(3):
        state, stats, _ = train_step(None, state, batch, 0.3, None)
      torch.cuda.synchronize()
      out.append(state.flat.clone())
  finally:
    train_utils._NT_DYNQ = old
  assert torch.equal(out[0], out[1])
