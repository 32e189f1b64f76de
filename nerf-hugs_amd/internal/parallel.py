"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The path shards by rays; the only exchange per step is one all-reduce of the flat
gradient buffer with the step's scalar stats in its tail (reference: jax.lax.pmean of grads and stats,
train_utils.py:457-459), and an all-gather of rendered chunks at eval (train_utils.py:559)."""
import torch
import torch.distributed as dist


def world_size():
  return dist.get_world_size() if dist.is_initialized() else 1


def rank():
  return dist.get_rank() if dist.is_initialized() else 0


def shard_batch(batch, rank_, world):
  """Rank's slice of a global batch along the leading (patch) axis -- whole P x P patches stay on one rank
  (RobustNeRF's box filter / patch vote are per patch, train_utils.py:284-310).  Raises like train.py:53-56
  when the batch does not divide."""
  from . import utils
  n = batch.rgb.shape[0]
  if n % world != 0:
    raise ValueError('Batch size must be divisible by the number of devices.')
  per = n // world
  return utils.tree_map(lambda x: x[rank_ * per:(rank_ + 1) * per], batch)


def allreduce_mean_(flat):
  """In-place mean over ranks of a flat buffer (sum here; callers that fuse the 1/world scale into the
  optimizer kernel pass scale=False)."""
  w = world_size()
  if w > 1:
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / w)
  return flat


def all_gather_cat(x):
  """Concatenate equally-shaped per-rank tensors along dim 0 (eval chunks)."""
  w = world_size()
  if w == 1:
    return x
  out = [torch.empty_like(x) for _ in range(w)]
  dist.all_gather(out, x.contiguous())
  return torch.cat(out, 0)
