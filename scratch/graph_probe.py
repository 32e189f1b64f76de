"""Feasibility probe (round 4): capture the existing train step in a hipGraph through torch.cuda.graph (scalars baked)
and compare replay time with the eager enqueue at 128..1024 rays.  Not a product path."""
import sys, os, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from nerf_hugs_amd.internal import configs, train_utils, random as hrandom

_keep = []
_OrigEvent = torch.cuda.Event
def _event(*a, **k):      # events must outlive the capture (a destroyed event that a captured wait still references: segfault in capture_end)
  e = _OrigEvent(*a, **k); _keep.append(e); return e
torch.cuda.Event = _event

class DummyStats(dict):
  def __init__(self, packed, build):
    super().__init__()

def run(rays):
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, bench.GIN)
  P = 16 if rays % 256 == 0 else 8
  config = configs.make_config(batch_size=rays)
  dev = torch.device('cuda', 0)
  model, state, _, train_step, _ = train_utils.setup_model(config, 20200823, compute_dtype='bf16', device=dev)
  batch = bench.synth_batch(rays // (P * P), P, 1000, dev)
  key = hrandom.PRNGKey(20200823, dev).clone()
  for _ in range(5):
    state, stats, key = train_step(key, state, batch, 0.5, None)
  torch.cuda.synchronize()
  def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3
  def eager():
    nonlocal state, key
    state, _, key = train_step(key, state, batch, 0.5, None)
  e_ms, e_host = timeit(eager)
  flat0 = state.flat.clone()
  # capture
  train_utils.LazyStats = DummyStats
  key_static = key.clone()
  g = torch.cuda.CUDAGraph()
  s = torch.cuda.Stream()
  s.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s, capture_error_mode='relaxed'):
      st2, _, newkey = train_step(key_static, state, batch, 0.5, None)
      key_static.copy_(newkey)
  torch.cuda.synchronize()
  g_ms, g_host = timeit(lambda: g.replay())
  ok = bool(torch.isfinite(state.flat).all()) and not torch.equal(flat0, state.flat)
  return dict(rays=rays, eager_ms=round(e_ms, 3), eager_host_ms=round(e_host, 3), graph_ms=round(g_ms, 3), graph_host_ms=round(g_host, 3), params_moved_and_finite=ok)

for r in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]:
  try:
    print(json.dumps(run(r)), flush=True)
  except Exception as e:
    import traceback; traceback.print_exc()
    print(json.dumps(dict(rays=r, error=repr(e)[:300])), flush=True)
