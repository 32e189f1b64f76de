"""The nerfacto path's fp16 mode (the reference's `enable_amp: True`, nerfacto/configs/phototourism_nerfacto_base.yml:3,
nerfacto/train.py:168,199-213): IEEE-half MFMA operands and activations (dtype code 2 of the C ABI), half copies of the
hash tables for the forward gathers, fp32 master parameters / accumulation, and torch.cuda.amp.GradScaler's dynamic loss
scale kept on the device.  Checked here: the half GEMMs against fp32 matmuls of the same half-rounded operands, the
half-table gather against the fp32 gather of the rounded table (bit-equal), a whole step against the fp32 oracle at half
precision, and the scaler + Adam sequence against torch.cuda.amp.GradScaler + torch.optim.Adam themselves."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = 'cuda'


def _L():
  from nerf_hugs_amd import _lib as L
  return L


@pytest.mark.parametrize('M,N,K,relu', [(256, 256, 256, 1), (1024, 128, 192, 0), (32768, 1024, 512, 1), (512, 384, 128, 1), (65536, 256, 256, 0)])
def test_half_gemm_nt_vs_fp32_matmul(M, N, K, relu):
  """C = relu?(A W^T + b) with half operands, fp32 accumulate, half output: every 256x256 / 256x128 / 128x128 / persistent
  kernel of hugs_gemm.hip compiled for dtype 2 (csrc/hugs_gemm_f16.hip)."""
  L = _L()
  g = torch.Generator(device=dev).manual_seed(M + N + K)
  A = torch.randn(M, K, generator=g, device=dev).half(); Wt = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).half()
  b = torch.randn(N, generator=g, device=dev)
  out = torch.empty(M, N, device=dev, dtype=torch.float16)
  L.call('hugs_gemm_nt', 2, M, N, K, 0, A, K, None, 0, Wt, K, b, None, 1, 0, relu, None, 0, None, None, out, N)
  ref = A.float() @ Wt.float().t() + b
  if relu:
    ref = ref.clamp_min(0)
  err = (out.float() - ref).abs()
  assert float((err / (ref.abs() + 1.0)).max()) < 1.5e-3          # half has 11 significant bits: 2^-11 = 4.9e-4 per rounding
  # the same operands through the bf16 build differ (8 significant bits): the two builds are really two formats
  out_b = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  L.call('hugs_gemm_nt', 1, M, N, K, 0, A.float().bfloat16(), K, None, 0, Wt.float().bfloat16(), K, b, None, 1, 0, relu, None, 0, None, None, out_b, N)
  assert float((out_b.float() - ref).abs().max()) > 2 * float(err.max())


@pytest.mark.parametrize('dt,K', [(2, 256), (2, 128), (1, 128)])
def test_gemm_mask_bits_and_tn(dt, K):
  """Relu mask bits written by a forward layer and consumed by the dX GEMM == the activation-mask form, for the persistent
  (K = 256: 8 K-stages) and the one-tile-per-workgroup (K = 128, the nerfacto field's first layers) 256 x 256 kernels, in
  both 16-bit formats; and the half TN weight-gradient GEMM."""
  L = _L()
  tdt = torch.float16 if dt == 2 else torch.bfloat16
  eps = 2.0 ** -11 if dt == 2 else 2.0 ** -8
  g = torch.Generator(device=dev).manual_seed(3 + K)
  M, N = 65536 + 256 * 8, 256                     # > 256 tiles
  A = torch.randn(M, K, generator=g, device=dev).to(tdt); Wt = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(tdt)
  b = torch.randn(N, generator=g, device=dev)
  Y = torch.empty(M, N, device=dev, dtype=tdt)
  bits = torch.zeros(L.lib().cdll.hugs_gemm_nt_bits_bytes(M, N) // 4, device=dev, dtype=torch.int32)
  L.call('hugs_gemm_nt_bits', dt, M, N, K, 0, A, K, None, 0, Wt, K, b, 1, None, None, Y, N, bits, None)
  ref = (A.float() @ Wt.float().t() + b).clamp_min(0)
  assert float(((Y.float() - ref).abs() / (ref + 1)).max()) < 3 * eps
  Y2 = torch.empty_like(Y)
  L.call('hugs_gemm_nt', dt, M, N, K, 0, A, K, None, 0, Wt, K, b, None, 1, 0, 1, None, 0, None, None, Y2, N)
  assert torch.equal(Y, Y2)                        # writing the bits does not change the output
  # dX[M, N] = (G W2^T) * (Y > 0) with G [M, K]: through the bit mask and through the activation mask
  G = torch.randn(M, K, generator=g, device=dev).to(tdt)
  W2 = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(tdt)
  dX1 = torch.empty(M, N, device=dev, dtype=tdt); dX2 = torch.empty_like(dX1)
  L.call('hugs_gemm_nt_bits', dt, M, N, K, 0, G, K, None, 0, W2, K, None, 0, None, None, dX1, N, None, bits)
  L.call('hugs_gemm_nt', dt, M, N, K, 0, G, K, None, 0, W2, K, None, None, 1, 0, 0, Y, N, None, None, dX2, N)
  assert torch.equal(dX1, dX2)
  refd = (G.float() @ W2.float().t()) * (Y > 0)
  assert float(((dX1.float() - refd).abs() / (refd.abs() + 1)).max()) < 4 * eps
  # dW = X^T G, db = colsum(G), fp32 out
  ns = 16
  ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N, ns) // 4, device=dev)
  dW, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
  Mt = 65536
  Gy = torch.randn(Mt, N, generator=g, device=dev).to(tdt)
  L.call('hugs_gemm_tn', dt, Mt, K, N, ns, A[:Mt], K, Gy, N, dW, db, ws)
  refw = A[:Mt].float().t() @ Gy.float()
  assert float((dW - refw).abs().max()) < 2e-3 * float(refw.abs().max())
  assert float((db - Gy.float().sum(0)).abs().max()) < 2e-3 * float(Gy.float().sum(0).abs().max()) + 1e-2


def test_half_table_gather_equals_fp32_gather_of_the_rounded_table():
  from nerf_hugs_amd.nerfacto.encodings import HashGrid
  L = _L()
  g = HashGrid(8, 2, 12, 16, None, 256, device=dev)
  gen = torch.Generator(device=dev).manual_seed(0)
  table = (torch.rand(g.n_entries, 2, generator=gen, device=dev) * 2 - 1)
  x = torch.rand(5000, 3, generator=gen, device=dev)
  o, r, s = g._tables()
  th = table.half()
  out_h = torch.empty(5000, 16, device=dev); out_f = torch.empty(5000, 16, device=dev)
  L.call('hugs_hashgrid_fwd_t', 5000, 8, 2, o, r, s, x, th, 2, 0, 16, out_h)
  L.call('hugs_hashgrid_fwd', 5000, 8, 2, o, r, s, x, th.float().contiguous(), 0, 16, out_f)
  assert torch.equal(out_h, out_f)                 # same fp32 interpolation, only the 16-bit loads differ
  out16 = torch.empty(5000, 16, device=dev, dtype=torch.float16)
  L.call('hugs_hashgrid_fwd_t', 5000, 8, 2, o, r, s, x, th, 2, 2, 16, out16)
  assert torch.equal(out16, out_f.half())
  # gradient rows in half: the table gradient equals the one from the same values in fp32
  d16 = torch.randn(5000, 16, generator=gen, device=dev).half()
  g1, g2 = torch.zeros_like(table), torch.zeros_like(table)
  L.call('hugs_hashgrid_bwd', 5000, 8, 2, o, r, s, x, d16, 2, 16, g1)
  L.call('hugs_hashgrid_bwd', 5000, 8, 2, o, r, s, x, d16.float().contiguous(), 0, 16, g2)
  assert float((g1 - g2).abs().max()) <= 1e-5 * float(g2.abs().max())       # (atomic order differs run to run)


SMALL = dict(num_levels=4, max_res=64, log2_hashmap_size=10, hidden_dim=16, geo_feat_dim=7, hidden_dim_color=16,
             num_proposal_samples_per_ray=(32, 16), num_nerf_samples_per_ray=8, opaque_background=True,
             use_appearance_embedding=True, appearance_embedding_dim=5, num_embedding=4, distortion_loss_mult=0.01,
             proposal_net_args_list=[dict(hidden_dim=8, log2_hashmap_size=9, num_levels=3, max_res=32)])


def _rays(N, seed):
  g = torch.Generator().manual_seed(seed)
  d = torch.randn(N, 3, generator=g); d = d / d.norm(dim=-1, keepdim=True)
  return dict(origin=torch.randn(N, 3, generator=g) * 0.3, direction=d * (0.8 + 0.4 * torch.rand(N, 1, generator=g)), viewdir=d,
              near=torch.full((N,), 0.05), far=torch.full((N,), 3.0), embed_idx=torch.randint(0, 4, (N,), generator=g).int(),
              bg_rgb=torch.ones(N, 3), rgb=torch.rand(N, 3, generator=g)), g


@pytest.mark.parametrize('variant', ['base', 'hanerf', 'prop_gemm'])
def test_fp16_step_vs_fp32_oracle(variant, monkeypatch):
  """Forward, losses and every leaf gradient of one fp16-mode step against the float32 oracle: rendered colour to 4e-3,
  losses to 1 %, gradients (divided by the loss scale the step applied) to 4 % of the leaf maximum -- half precision, not
  the 1e-4 of the fp32 parity mode -- and scale x gradient is what the buffer holds."""
  from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
  from oracle import nerfacto_ref as NF
  kw = dict(SMALL)
  if variant == 'hanerf':
    kw.update(transient_type='hanerf', use_transient_embedding=True, transient_embedding_dim=4, num_levels_implicit=3, base_res_implicit=4,
              max_res_implicit=32, log2_hashmap_size_implicit=8, hidden_dim_implicit=16, features_per_level_implicit=2)
  if variant == 'prop_gemm':
    monkeypatch.setenv('HUGS_NF_FUSED_PROP', '0')
  ocfg = NF.Cfg(**kw)
  P = NF.init_params(ocfg, 3)
  for k in P:
    if isinstance(P[k], dict) and 'table' in P[k]:
      P[k]['table'] = P[k]['table'] * 3e3
  model = NerfactoModel(NerfactoConfig(**kw), compute_dtype='fp16')
  model.load_params(P)
  N = 128
  b, g = _rays(N, 5)
  if variant == 'hanerf':
    b['coord'] = torch.rand(N, 2, generator=g)
  u01 = [torch.rand(N, generator=g) for _ in range(3)]
  leaves = []
  for grp in P.values():
    for v in (grp.values() if isinstance(grp, dict) else [grp]):
      v.requires_grad_(True); leaves.append(v)
  orays = {k: (v[:, None] if v.dim() == 1 and k in ('near', 'far', 'embed_idx') else v) for k, v in b.items()}
  out = NF.forward_rays(ocfg, P, orays, 300, [u[:, None] for u in u01])
  loss, info = NF.loss_fn(ocfg, out, b['rgb'], torch.zeros(N)[:, None], 1.0, curr_step=300)
  loss.backward()
  gb = {k: v.to(dev) for k, v in b.items()}
  assert model.loss_scale() == 65536.0
  res = model.train_step(gb, curr_step=300, u01=[u.to(dev) for u in u01], apply_update=False)
  torch.cuda.synchronize()
  lv = res['levels']
  np.testing.assert_allclose(lv[-1]['rgb_out'].cpu().numpy(), out['rgb'].detach().numpy(), rtol=0, atol=4e-3)
  st = res['stats'].cpu().numpy()
  assert abs(st[1] - float(info['rgb_loss'])) <= 1e-2 * abs(float(info['rgb_loss']))
  assert abs(st[2] + st[3] - float(info['interlevel_loss'])) <= 2e-2 * float(info['interlevel_loss']) + 1e-9
  mg = model.grads()
  assert bool(torch.isfinite(model.grad).all())
  for name, grp in P.items():
    for k, v in (grp.items() if isinstance(grp, dict) else [(None, grp)]):
      mine = (mg[name][k] if k else mg[name]).cpu().double() / 65536.0
      ref = v.grad.double()
      sc = float(ref.abs().max())
      assert sc > 0, (name, k)
      err = float((mine - ref).abs().max()) / sc
      assert err < 4e-2, f'{variant} grad {name}/{k}: rel err {err:.2e} (max |g| {sc:.2e})'


def test_scaler_and_adam_sequence_equals_torch_amp():
  """scaler.step(optimizer) / scaler.update() / scheduler.step() over 14 steps with overflows on steps 0, 1 and 9, no
  proposal update on steps 4 and 5 (their .grad is None: skipped by Adam AND by the inf check), growth_interval 3: the scale
  trajectory, the parameters and both moments equal torch.cuda.amp.GradScaler + torch.optim.Adam run on the same numbers."""
  from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
  model = NerfactoModel(NerfactoConfig(**dict(SMALL, lr_init=1e-2, warmup_steps=4, num_steps=20)), compute_dtype='fp16', seed=2)
  model.amp_opts['growth_interval'] = 3
  names = [n for n in model.group_order if model.groups[n][1] > model.groups[n][0]]
  tp = {n: torch.nn.Parameter(model.flat[model.groups[n][0]:model.groups[n][1]].clone()) for n in names}
  o = model._opt()
  opt = torch.optim.Adam([{'params': [tp[n]], 'lr': o['lr_init']} for n in names], betas=o['betas'], eps=o['eps'])
  scaler = torch.amp.GradScaler('cuda', growth_interval=3)
  scaler.scale(torch.zeros(1, device=dev))          # (creates the scale tensor, as the first scaler.scale(loss) does)
  gen = torch.Generator(device=dev).manual_seed(7)
  scales = []
  for step in range(14):
    prop_on = step not in (4, 5)
    S = float(scaler.get_scale())
    assert model.loss_scale() == S, (step, model.loss_scale(), S)
    scales.append(S)
    for pg in opt.param_groups:
      pg['lr'] = model.lr(step)                      # LambdaLR factor of the `step`-th scheduler step (train.py:214)
    model.grad.zero_()
    for n in names:
      lo, hi = model.groups[n]
      if n == 'proposal' and not prop_on:
        tp[n].grad = None
        continue
      gtrue = torch.randn(hi - lo, generator=gen, device=dev) * 1e-3
      gs = gtrue * S
      if step in (0, 9) and n == 'field':
        gs[5] = float('inf')
      if step == 1 and n == names[-1]:
        gs[0] = float('nan')
      model.grad[lo:hi] = gs
      tp[n].grad = gs.clone()
    scaler.step(opt)
    scaler.update()
    model.apply_gradients(prop_on)
  assert scales[2] == 16384.0 and max(scales[3:]) > 16384.0 and any(b < a for a, b in zip(scales[3:], scales[4:])), scales      # backoff x2, growth, backoff again
  for n in names:
    lo, hi = model.groups[n]
    np.testing.assert_allclose(model.flat[lo:hi].cpu().numpy(), tp[n].detach().cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg=n)
    stt = opt.state[tp[n]]
    np.testing.assert_allclose(model.m[lo:hi].cpu().numpy(), stt['exp_avg'].cpu().numpy(), rtol=2e-6, atol=1e-9, err_msg=n)
    np.testing.assert_allclose(model.v[lo:hi].cpu().numpy(), stt['exp_avg_sq'].cpu().numpy(), rtol=2e-6, atol=1e-12, err_msg=n)
    gi = model.group_order.index(n)
    assert float(model.amp_counts[gi]) == float(stt['step']), (n, float(model.amp_counts[gi]), float(stt['step']))
  assert model.loss_scale() == float(scaler.get_scale())


def test_fp16_training_reduces_the_loss_recovers_from_overflow_and_is_reproducible():
  from nerf_hugs_amd.nerfacto.model import NerfactoConfig, NerfactoModel
  losses, scales = [], []
  for rep in range(2):
    model = NerfactoModel(NerfactoConfig(**dict(SMALL, lr_init=5e-3, warmup_steps=5)), compute_dtype='fp16', seed=11)
    model.amp_state[0] = 2.0 ** 34                   # start far too high: the first steps overflow in half and are skipped
    b, g = _rays(256, 9)
    b = {k: v.to(dev) for k, v in b.items()}
    gen = torch.Generator(device=dev).manual_seed(1)
    run = []
    p0 = model.flat.clone()
    for i in range(80):
      u01 = [torch.rand(256, generator=gen, device=dev) for _ in range(3)]
      res = model.train_step(b, u01=u01)
      run.append(float(res['stats'][1]))
      if i == 0:
        assert torch.equal(model.flat, p0) and model.loss_scale() == 2.0 ** 33      # overflow: step skipped, scale halved
    assert all(np.isfinite(run)) and bool(torch.isfinite(model.flat).all())
    losses.append(run); scales.append(model.loss_scale())
  assert scales[0] < 2.0 ** 30 and scales[0] == scales[1], scales
  assert np.mean(losses[0][-5:]) < 0.7 * np.mean(losses[0][:5]), losses[0]
  np.testing.assert_allclose(losses[0], losses[1], rtol=2e-2)
