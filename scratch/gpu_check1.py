"""Early GPU check: stepfun bit-exactness, encode, GEMM NT/TN (bf16 + fp32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_hugs_amd import _lib as L
from oracle import cstepfun as C, torch_ref as R
dev = 'cuda'
torch.manual_seed(0)
# --- exp/log bit exact
x = torch.cat([torch.linspace(-104, 89, 200001), torch.rand(100000) * 1e-30, torch.rand(100000)]).float()
ye = torch.empty_like(x, device=dev); yl = torch.empty_like(x, device=dev)
L.call('hugs_test_explog', x.to(dev), x.numel(), ye, yl)
ce, cl = C.expf(x.numpy()), C.logf(x.numpy())
print('expf bit-exact:', np.array_equal(ye.cpu().numpy().view(np.uint32), ce.view(np.uint32)),
      'logf bit-exact:', np.array_equal(yl.cpu().numpy().view(np.uint32), cl.view(np.uint32)))
# --- level sample
rng = np.random.default_rng(0)
def run_level(N, n_prev, ns, dil, raydist, jitter):
    t = np.sort(rng.uniform(0, 1, (N, n_prev + 1)).astype(np.float32), -1) if n_prev > 1 else np.tile(np.array([[0., 1.]], np.float32), (N, 1))
    w = rng.uniform(0, 1, (N, n_prev)).astype(np.float32) ** 4
    w[rng.uniform(size=w.shape) < 0.1] = 0
    w /= np.maximum(w.sum(-1, keepdims=True), 1e-9)
    ub, mj = R.sample_u_base(ns, jitter)
    jit = (rng.random(N, dtype=np.float32) * np.float32(mj)).astype(np.float32) if jitter else None
    near = rng.uniform(0.05, 0.3, N).astype(np.float32); far = np.full(N, 1e6 if raydist else 1.2, np.float32)
    sd_o, td_o, idx_o = C.level_sample(t, w, dil is not None, dil or 0., 0., 1., 0.7, 0., ub, jit, raydist, near, far)
    g = lambda a: torch.from_numpy(a).to(dev)
    sd = torch.empty(N, ns + 1, device=dev); td = torch.empty(N, ns + 1, device=dev); idx = torch.empty(N, ns, dtype=torch.int32, device=dev)
    L.call('hugs_level_sample_fwd', N, g(t), g(w), n_prev, int(dil is not None), dil or 0., 0., 1., 0.7, 0., g(ub),
           g(jit) if jit is not None else None, 1, ns, raydist, g(near), g(far), sd, td, idx)
    torch.cuda.synchronize()
    ok = (np.array_equal(sd.cpu().numpy().view(np.uint32), sd_o.view(np.uint32)), np.array_equal(td.cpu().numpy().view(np.uint32), td_o.view(np.uint32)), np.array_equal(idx.cpu().numpy(), idx_o))
    print(f'level N={N} n_prev={n_prev} ns={ns} dil={dil} rd={raydist} jit={jitter}: sdist/tdist/idx bit-exact = {ok}', 'maxdiff', np.abs(sd.cpu().numpy() - sd_o).max())
run_level(1000, 1, 64, None, 0, True)
run_level(1000, 64, 128, 0.0103125, 0, True)
run_level(1000, 64, 64, 0.0103125, 1, False)
run_level(999, 64, 32, 0.00262, 1, True)
run_level(513, 85, 256, 0.003, 0, True)
# --- encode
N, S = 64, 128
basis = torch.tensor(R.generate_basis('icosahedron', 2).T.copy(), dtype=torch.float32)
o = torch.randn(N, 3) * 0.5; d = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1) * (0.8 + 0.4 * torch.rand(N, 1))
radii = 5e-4 + 1.5e-3 * torch.rand(N, 1)
td = torch.sort(torch.rand(N, S + 1) * 3 + 0.1, -1).values
for warp in (0, 1):
    means, covs = R.cast_rays(td, o, d, radii)
    if warp: means, covs = R.contract_track_linearize(means, covs)
    lm, lv = R.lift_and_diagonalize(means, covs, basis)
    ref = R.integrated_pos_enc(lm, lv, 0, 12).reshape(N * S, 504)
    out = torch.empty(N * S, 512, device=dev)
    L.call('hugs_cast_ipe_fwd', N, S, td.to(dev), o.to(dev), d.to(dev), radii.to(dev), basis.to(dev), 0, warp, 12, 0, out)
    outb = torch.empty(N * S, 512, device=dev, dtype=torch.bfloat16)
    L.call('hugs_cast_ipe_fwd', N, S, td.to(dev), o.to(dev), d.to(dev), radii.to(dev), basis.to(dev), 0, warp, 12, 1, outb)
    err = (out.cpu()[:, :504] - ref).abs().max().item(); errb = (outb.float().cpu()[:, :504] - ref).abs().max().item()
    print(f'ipe warp={warp}: max abs err fp32 {err:.3e} bf16 {errb:.3e}; pad zero: {out[:, 504:].abs().max().item()}')
# --- gemm
def gemm_check(dtype, M, N_, K1, K2, relu=1, mask=False, r1=False, rowbias=False):
    tdt = torch.bfloat16 if dtype else torch.float32
    A1 = torch.randn(M, K1, device=dev).to(tdt); A2 = torch.randn(M, K2, device=dev).to(tdt) if K2 else None
    Bt = (torch.randn(N_, K1 + K2, device=dev) / (K1 + K2) ** 0.5).to(tdt)
    bias = torch.randn(N_, device=dev)
    rb = torch.randn(M // 64, N_, device=dev) if rowbias else None
    mk = torch.randn(M, N_, device=dev).to(tdt) if mask else None
    rr = torch.randn(M, device=dev) if r1 else None; rc = torch.randn(N_, device=dev) if r1 else None
    out = torch.empty(M, N_, device=dev, dtype=tdt)
    L.call('hugs_gemm_nt', dtype, M, N_, K1, K2, A1, K1, A2, K2, Bt, K1 + K2, bias, rb, 64, N_, relu, mk, N_, rr, rc, out, N_)
    A = torch.cat([A1, A2], 1) if K2 else A1
    ref = A.double() @ Bt.double().T + bias.double()
    if rowbias: ref += rb.double().repeat_interleave(64, 0)
    if r1: ref += rr.double()[:, None] * rc.double()[None]
    if relu: ref = ref.clamp(min=0)
    if mask: ref = ref * (mk.double() > 0)
    err = (out.double() - ref).abs().max().item(); scale = ref.abs().max().item()
    print(f'gemm_nt dtype={dtype} M={M} N={N_} K={K1}+{K2} relu={relu} mask={mask} r1={r1} rb={rowbias}: max err {err:.3e} (scale {scale:.2f})')
for dt in (0, 1):
    gemm_check(dt, 256, 128, 64 if dt else 64, 0)
    gemm_check(dt, 1024, 1024, 1024, 512)
    gemm_check(dt, 640, 256, 512, 0, relu=0, mask=True, r1=True)
    gemm_check(dt, 384, 128, 256, 0, rowbias=True)
def tn_check(dtype, Mr, Kc, N_, ns):
    tdt = torch.bfloat16 if dtype else torch.float32
    X = torch.randn(Mr, Kc, device=dev).to(tdt); G = torch.randn(Mr, N_, device=dev).to(tdt)
    dW = torch.empty(Kc, N_, device=dev); db = torch.empty(N_, device=dev)
    ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(Kc, N_, ns) // 4, device=dev)
    L.call('hugs_gemm_tn', dtype, Mr, Kc, N_, ns, X, Kc, G, N_, dW, db, ws)
    ref = X.double().T @ G.double(); rb = G.double().sum(0)
    print(f'gemm_tn dtype={dtype} rows={Mr} Kc={Kc} N={N_} split={ns}: dW err {(dW.double()-ref).abs().max().item():.3e} (scale {ref.abs().max().item():.1f}) db err {(db.double()-rb).abs().max().item():.3e}')
for dt in (0, 1):
    tn_check(dt, 1024, 128, 128, 1)
    tn_check(dt, 4096, 512, 256, 4)
    tn_check(dt, 8192, 1024, 1024, 8)
# --- quick perf of the big GEMM
M, N_, K = 131072, 1024, 1024
A = torch.randn(M, K, device=dev).bfloat16(); Bt = (torch.randn(N_, K, device=dev) / 32).bfloat16(); bias = torch.zeros(N_, device=dev)
out = torch.empty(M, N_, device=dev, dtype=torch.bfloat16)
for it in range(3): L.call('hugs_gemm_nt', 1, M, N_, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N_)
torch.cuda.synchronize(); t0 = time.time()
for it in range(10): L.call('hugs_gemm_nt', 1, M, N_, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, 0, None, None, out, N_)
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f'bf16 NT 131072x1024x1024: {dt*1e3:.3f} ms  {2*M*N_*K/dt/1e12:.1f} TF')
G = torch.randn(M, N_, device=dev).bfloat16(); dW = torch.empty(K, N_, device=dev); db = torch.empty(N_, device=dev)
ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N_, 8) // 4, device=dev)
for it in range(3): L.call('hugs_gemm_tn', 1, M, K, N_, 8, A, K, G, N_, dW, db, ws)
torch.cuda.synchronize(); t0 = time.time()
for it in range(10): L.call('hugs_gemm_tn', 1, M, K, N_, 8, A, K, G, N_, dW, db, ws)
torch.cuda.synchronize(); dt = (time.time() - t0) / 10
print(f'bf16 TN 131072 rows 1024x1024 split 8: {dt*1e3:.3f} ms  {2*M*N_*K/dt/1e12:.1f} TF')
Af = A.float(); Btf = Bt.float(); outf = torch.empty(M, N_, device=dev)
for it in range(2): L.call('hugs_gemm_nt', 0, M, N_, K, 0, Af, K, None, 0, Btf, K, bias, None, 1, 0, 1, None, 0, None, None, outf, N_)
torch.cuda.synchronize(); t0 = time.time()
for it in range(5): L.call('hugs_gemm_nt', 0, M, N_, K, 0, Af, K, None, 0, Btf, K, bias, None, 1, 0, 1, None, 0, None, None, outf, N_)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f'fp32 NT 131072x1024x1024: {dt*1e3:.3f} ms  {2*M*N_*K/dt/1e12:.1f} TF')
