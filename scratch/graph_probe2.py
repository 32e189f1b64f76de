"""Bisect which part of the step breaks hipGraph capture (each mode in its own process)."""
import sys, os, time, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = ['single:128', 'single:256', 'single:512', 'single:1024', 'lanes03:1024', 'lanes03:128']
if len(sys.argv) == 1:
  for m in MODES:
    env = dict(os.environ)
    env['HUGS_SIDE_LANES'] = '9' if m.startswith('single') else '0,3'
    env['PROBE_RAYS'] = m.split(':')[1]
    r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, m], env=env, capture_output=True, text=True)
    tail = (r.stdout + r.stderr).strip().splitlines()[-6:]
    print(m, 'rc', r.returncode, '|', ' / '.join(tail)[-700:], flush=True)
  sys.exit(0)
mode = sys.argv[1]
import numpy as np, torch
import bench
from nerf_hugs_amd.internal import configs, train_utils, random as hrandom, models as M
_keep = []
_OrigEvent = torch.cuda.Event
def _event(*a, **k):
  e = _OrigEvent(*a, **k); _keep.append(e); return e
torch.cuda.Event = _event
class DummyStats(dict):
  def __init__(self, packed, build): super().__init__()
train_utils.LazyStats = DummyStats
dev = torch.device('cuda', 0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
  configs.clear_config(); configs.parse_config_files_and_bindings(None, bench.GIN)
  RAYS = int(os.environ.get('PROBE_RAYS', '256'))
  config = configs.make_config(batch_size=RAYS)
  model, state, _, train_step, _ = train_utils.setup_model(config, 20200823, compute_dtype='bf16', device=dev)
  P_ = 16 if RAYS % 256 == 0 else 8
  batch = bench.synth_batch(RAYS // (P_ * P_), P_, 1000, dev)
  key = hrandom.PRNGKey(20200823, dev).clone()
  eng = model.engine(dev)
  rays = M.rays_to_dict(batch.rays, dev)
  x = torch.zeros(1024, device=dev); y = torch.zeros(1024, device=dev)
  s2 = torch.cuda.Stream()
  def body():
    global state, key
    if mode == 'torch_only':
      x.add_(1.0); y.copy_(x)
    elif mode == 'forkjoin_torch':
      x.add_(1.0)
      e = torch.cuda.Event(); e.record()
      with torch.cuda.stream(s2):
        s2.wait_event(e); y.add_(2.0)
        e2 = torch.cuda.Event(); e2.record(s2)
      torch.cuda.current_stream().wait_event(e2)
      x.add_(y)
    elif mode == 'cast':
      eng.refresh_weights(state.flat)
    elif mode == 'forward':
      eng.forward(state.flat, rays, 0.5, None, False)
    else:
      state, _, nk = train_step(key, state, batch, 0.5, None)
      key.copy_(nk)
  for _ in range(3): body()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20): body()
  the = time.perf_counter() - t0
  torch.cuda.synchronize()
  print('eager ms', round((time.perf_counter() - t0) / 20 * 1e3, 3), 'host ms', round(the / 20 * 1e3, 3))
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=s, capture_error_mode='relaxed'):
    body()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(20): g.replay()
  th = time.perf_counter() - t0
  torch.cuda.synchronize()
  print('OK', mode, 'replay ms', round((time.perf_counter() - t0) / 20 * 1e3, 3), 'host ms', round(th / 20 * 1e3, 3), 'finite', bool(torch.isfinite(state.flat).all()))
