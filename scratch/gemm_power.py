"""Is the trunk GEMM limited by the chip's power management?  Same launch, same shape, different operand DATA: zeros, a
constant, small integers, N(0,1).  A data-dependent duration at identical instruction streams is switching power -> clock."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nerf_hugs_amd import _lib as L
dev = 'cuda'
M, N, K = 131072, 1024, 1024

def fill(kind, shape, scale=1.0):
  if kind == 'zeros': return torch.zeros(shape, device=dev).bfloat16()
  if kind == 'ones': return torch.full(shape, 1.0, device=dev).bfloat16()
  if kind == 'smallint': return torch.randint(-3, 4, shape, device=dev).float().bfloat16()
  return (torch.randn(shape, device=dev) * scale).bfloat16()

smi = []
def sample(stop):
  while not stop.is_set():
    try:
      o = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True, timeout=5).stdout.strip().splitlines()[-1]
      smi.append(o)
    except Exception as e:
      smi.append(repr(e))
    time.sleep(0.05)

def perf(kind, nt=True, reps=200):
  A = fill(kind, (M, K)); Bt = fill(kind, (N, K), 1 / 32); bias = torch.zeros(N, device=dev)
  out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
  if nt:
    f = lambda: L.call('hugs_gemm_nt_tiles', 0, 1, M, N, K, 0, A, K, None, 0, Bt, K, bias, None, 1, 0, 1, None, N, None, None, out, N)
  else:
    G = fill(kind, (M, N)); dW = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
    ws = torch.empty(L.lib().cdll.hugs_gemm_tn_ws_bytes(K, N, 16) // 4, device=dev)
    f = lambda: L.call('hugs_gemm_tn', 1, M, K, N, 16, A, K, G, N, dW, db, ws)
  for _ in range(30): f()
  torch.cuda.synchronize()
  smi.clear(); stop = threading.Event(); th = threading.Thread(target=sample, args=(stop,)); th.start()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize()
  stop.set(); th.join()
  dt = e0.elapsed_time(e1) / reps * 1e-3
  mid = smi[len(smi) // 2] if smi else ''
  print(f'{"NT fwd" if nt else "TN dW "} {kind:9s}: {dt*1e6:7.1f} us {2*M*N*K/dt/1e12:6.0f} TF   smi mid-run: {mid}', flush=True)

print(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--csv'], capture_output=True, text=True).stdout.strip().splitlines()[0])
for rep in range(2):
  for kind in ('zeros', 'ones', 'smallint', 'randn'):
    perf(kind, True)
  for kind in ('zeros', 'smallint', 'randn'):
    perf(kind, False)
