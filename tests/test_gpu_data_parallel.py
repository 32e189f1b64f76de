"""Data-parallel correctness on one GPU: two ranks (gloo backend, both on cuda:0) each take half of a batch;
with deterministic sampling and the base MSE loss (per-device normalisers are equal-size means) one DP step must
reproduce the single-process step on the whole batch up to float reassociation of the gradient sum."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GIN = ["Config.patch_size = 8", "Config.data_loss_type = 'mse'", "Config.distortion_loss_mult = 0.01",
       "Config.randomized = False", "Model.opaque_background = True", "Model.num_levels = 3", "PropMLP.net_depth = 4",
       "PropMLP.net_width = 128", "PropMLP.disable_rgb = True", "NerfMLP.net_depth = 8", "NerfMLP.net_width = 128"]


HANERF = GIN + ["Config.transient_type = 'hanerf'", "Model.num_transient_features = 16", "Model.num_glo_features = 4",
                "NerfMLP.bottleneck_width = 128"]


def _step(rank, world, port, out_dir, gin, backend='gloo', nsteps=1, graph='0', dtype='fp32', finetune=0):
  import sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import configs, train_utils, parallel
  torch.cuda.set_device(rank if backend == 'nccl' else 0)
  if world > 1:
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    if backend == 'nccl':           # RCCL: one GPU per rank
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    else:
      dist.init_process_group('gloo', rank=rank, world_size=world)
  train_utils._STEP_GRAPH = graph
  configs.clear_config()
  configs.parse_config_files_and_bindings(None, gin)
  config = configs.make_config()
  model, state, _, train_step, _ = train_utils.setup_model(config, 3, compute_dtype=dtype,
                                                           device=torch.device('cuda', torch.cuda.current_device()))
  batch = H.synth_rays(4, 8, 5)
  if world > 1:
    batch = parallel.shard_batch(batch, rank, world)
  for i in range(nsteps):
    state, stats, _ = train_step(None, state, batch, 0.4 + 0.01 * i, None)
  torch.cuda.synchronize()
  assert train_step.graph_active() == (graph == '1' and nsteps > 2)
  if finetune:          # train.py:97-109: the embedding-only stage behind the training steps, same sharding
    state, ftrain, _ = train_utils.setup_finetune_model(config, model, state)
    for i in range(finetune):
      state, stats, _ = ftrain(None, state, batch, 1.0, None)
    torch.cuda.synchronize()
  if rank == 0:
    torch.save({'flat': state.flat.cpu(), 'loss': float(stats['loss']), 'mses': stats['mses']}, os.path.join(out_dir, f'w{world}.pt'))
  if world > 1:
    dist.destroy_process_group()


def _compare(tmp_path, gin, backend, nsteps=1, graph='0', dtype='fp32', finetune=0):
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  mp.spawn(_step, args=(1, port, str(tmp_path), gin, backend, nsteps, graph, dtype, finetune), nprocs=1, join=True)
  mp.spawn(_step, args=(2, port, str(tmp_path), gin, backend, nsteps, graph, dtype, finetune), nprocs=2, join=True)
  a = torch.load(tmp_path / 'w1.pt'); b = torch.load(tmp_path / 'w2.pt')
  from tests import hugs_testlib as H
  from nerf_hugs_amd.internal import configs, models
  configs.clear_config(); configs.parse_config_files_and_bindings(None, gin)
  m = models.Model(configs.make_config())
  init = m.init(3, 'cpu')
  da, db = a['flat'] - init, b['flat'] - init
  assert float(da.abs().max()) > 0
  # clipped-Adam's first update is ~ lr * g/(|g|+eps): compare the updates leaf by leaf
  for lf in m.layout.leaves:
    ua, ub = m.layout.view(da, lf['path']), m.layout.view(db, lf['path'])
    # (several steps: Adam turns a gradient that differs in its last bits into an update that differs by ~lr * relative error,
    # and the reassociated gradient sums of the two shardings drift apart step by step: 5e-3 of the largest update per leaf)
    if nsteps == 1 and not finetune:
      assert float((ua - ub).abs().max()) <= 2e-3 * float(ua.abs().max()) + 3e-8, lf['path']
    else:      # a population statement: one ReLU decision that flips in step 2 moves single entries by a sample's contribution
      sc = float(ua.abs().max())
      bad = int(((ua - ub).abs() > 5e-3 * sc + 3e-8).sum())
      worst = float((ua - ub).abs().max()) / max(sc, 1e-30)
      assert bad <= max(2, 5e-3 * ua.numel()) and worst < 1e-1, (lf["path"], bad, ua.numel(), worst)
  assert abs(a['loss'] / b['loss'] - 1) < 1e-4      # pmean of per-shard losses == the full-batch loss here


# 256-wide nets: the trunk weight gradients take the batched launch (hugs_gemm_tn_batch), whose whole gradient range is released
# to ONE bucket behind it (the 128-wide nets above keep the per-layer launches and buckets)
GIN256 = [g for g in GIN if 'net_width' not in g] + ["PropMLP.net_width = 256", "NerfMLP.net_width = 256"]


NERFW = GIN + ["Config.transient_type = 'nerfw'", "Model.num_transient_features = 16", "Model.num_glo_features = 4",
               "NerfMLP.bottleneck_width = 128"]
NOVIEW = GIN + ["Model.use_viewdirs = False", "Model.num_glo_features = 4"]


@pytest.mark.parametrize('gin', [GIN, HANERF, NERFW, NOVIEW], ids=['base', 'hanerf', 'nerfw', 'no_viewdirs'])
def test_two_rank_step_equals_single_process(tmp_path, gin):
  _compare(tmp_path, gin, 'gloo')


def test_two_rank_finetune_stage_equals_single_process(tmp_path):
  """One training step, then two steps of the finetune stage (every gradient range goes through the un-bucketed all-reduce there; the
  coarse data loss keeps the proposal MLP in the backward pass), two ranks against one process."""
  gin = NERFW + ["Config.finetune_enable = True", "PropMLP.disable_rgb = False", "PropMLP.bottleneck_width = 128", "Config.data_coarse_loss_mult = 0.2"]
  _compare(tmp_path, gin, 'gloo', finetune=2)


def test_two_rank_step_equals_single_process_bf16_batched_dw(tmp_path):
  """bf16, 256-wide nets: the batched weight-gradient launch and its single trunk bucket.  Per-ray arithmetic does not depend
  on the sharding (same kernels row by row); only the fp32 reduction order of the weight gradients does."""
  _compare(tmp_path, GIN256, 'gloo', dtype='bf16')


def test_two_rank_graph_steps_bf16_batched_dw(tmp_path):
  """The captured data-parallel form in the shipped precision with the batched weight-gradient launch (round 5: the step's last launch
  publishes stats / key from the SECOND graph)."""
  _compare(tmp_path, GIN256, 'gloo', nsteps=4, graph='1', dtype='bf16')


def test_two_rank_graph_steps_equal_single_process(tmp_path):
  """The captured form of the data-parallel step (two hipGraphs around one all-reduce of the whole gradient buffer,
  train_utils.create_train_step) over four steps -- two eager, the capture, one replay -- against the single-process graph."""
  _compare(tmp_path, GIN, 'gloo', nsteps=4, graph='1')


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL needs one GPU per rank: runs on multi-GPU nodes only')
def test_two_rank_step_equals_single_process_rccl(tmp_path):
  """The same equality through backend 'nccl' (= RCCL over xGMI): the async all-reduce of the NerfMLP gradient
  segment overlapping the proposal backward, the side-stream dW GEMMs and RCCL's own stream in one step."""
  _compare(tmp_path, GIN, 'nccl')


@pytest.mark.parametrize('form', ['captured', 'eager_bucketed'])
def test_bench_two_ranks_on_one_gpu_prints_the_contract_line(form):
  """bench.py's N > 1 path end to end as the driver launches it (torch.distributed.run, one rank per process), two ranks on ONE
  GPU over gloo (HUGS_FORCE_DEVICE / HUGS_DIST_BACKEND test hooks): exactly one JSON line from rank 0 with the contract's keys,
  whole-job rays/s, and the exposed all-reduce time next to it -- for BOTH forms of the data-parallel step (round 6): the captured
  one (two hipGraphs around one all-reduce of the whole gradient buffer: the all-reduce never shares the chip with the persistent
  one-workgroup-per-CU GEMMs) and the eager one (buckets issued under the backward pass), so that the first real SCALE run can be
  read against either."""
  import json
  import os
  import socket
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, HUGS_FORCE_DEVICE='0', HUGS_DIST_BACKEND='gloo')
  out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '3',
                        '--min-time', '0', '--batch-pool', '4', '--step-graph', '1' if form == 'captured' else '0'],
                       capture_output=True, text=True, env=env, timeout=600, cwd=root)
  assert out.returncode == 0, out.stderr[-3000:]
  lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, out.stdout[-2000:]
  d = json.loads(lines[0])
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'fixed_batch', 'allreduce_exposed_ms_per_step', 'allreduce_form'):
    assert k in d, k
  assert d['n_gpus'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak' and d['config']['global_batch'] == 2048
  assert abs(d['value'] - 2048 * 4 / (d['ms_per_step'] * 4 * 1e-3)) < 1e-3 * d['value']      # whole-job rays/s
  assert d['allreduce_exposed_ms_per_step'] is not None and d['allreduce_exposed_ms_per_step'] >= 0
  assert d['step_graph'] == (form == 'captured')
  assert ('two hipGraphs' in d['allreduce_form']) == (form == 'captured')
  print(f"two ranks on one GPU over gloo, {form}: {d['ms_per_step']} ms/step, allreduce_exposed_ms_per_step {d['allreduce_exposed_ms_per_step']}")
  assert 'cpu_baseline' not in d      # rank 0 at N = 1 only
