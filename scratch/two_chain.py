"""Does running TWO independent half-batch train steps on two streams (offset in time) beat one full-batch step?  Two model
instances (same config, 512 rays each) stepped alternately from one thread, each inside its own stream; aggregate rays/s
against one instance with 1024 rays.  (The cheap way to find out whether pipelining two half batches inside one step would
hide the ~1 ms of latency-bound small kernels behind the other half's GEMMs.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as B
from nerf_hugs_amd.internal import configs, train_utils, random as hrandom
dev = torch.device('cuda:0')

def make(rays):
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, B.GIN)
  config = configs.make_config(batch_size=rays)
  model, state, _, train_step, _ = train_utils.setup_model(config, 20200823, compute_dtype='bf16', device=dev)
  batch = B.synth_batch(rays // 256, 16, 1000, dev)
  gen = hrandom.PRNGKey(20200823, dev)
  return dict(state=state, step=train_step, batch=batch, gen=gen)

def run_one(m):
  m['state'], st, m['gen'] = m['step'](m['gen'], m['state'], m['batch'], 0.5, None)

def timeit(fn, n):
  for _ in range(8): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

full = make(1024)
t_full = timeit(lambda: run_one(full), 60)
print(f'one chain, 1024 rays: {t_full*1e3:.3f} ms/step = {1024/t_full:.0f} rays/s', flush=True)
half = make(512)
t_half = timeit(lambda: run_one(half), 60)
print(f'one chain,  512 rays: {t_half*1e3:.3f} ms/step = {512/t_half:.0f} rays/s', flush=True)
a, b = make(512), make(512)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def both():
  with torch.cuda.stream(sa): run_one(a)
  with torch.cuda.stream(sb): run_one(b)
t2 = timeit(both, 60)
print(f'two chains, 2 x 512 rays on two streams: {t2*1e3:.3f} ms per pair = {1024/t2:.0f} rays/s', flush=True)
