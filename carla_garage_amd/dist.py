"""Data-parallel exchange of the MI355X path (team_code/train.py:361-365,516-520): one process per GPU, one flat fp32
gradient arena, one all-reduce per step.  Backend 'nccl' is RCCL over xGMI on ROCm; the same code runs on 'gloo' CPU
tensors, which is how tests/test_dist.py covers it without GPUs."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
  """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns (rank, local_rank, world)."""
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'), init_method='env://')
  return rank, local_rank, world


def world_size(group=None):
  return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def exchange_enabled(group=None):
  """True when the gradient collectives are really issued: more than one rank, or a process group of ONE rank with
  TFPP_FORCE_COLLECTIVES=1 (a 1-GPU box then runs the complete RCCL code path -- launch, stream ordering, the two-graph step
  with the all-reduce between the replays -- which is how tests/test_dist_gpu.py covers it without a second GPU)."""
  if not (dist.is_available() and dist.is_initialized()):
    return False
  return dist.get_world_size(group) > 1 or os.environ.get('TFPP_FORCE_COLLECTIVES', '0') == '1'


def assert_same_on_every_rank(value, what, device, group=None):
  """Raises on EVERY rank when the ranks hold different values of the 63-bit integer ``value`` (one tiny MIN / MAX all-reduce pair): used for
  things that must agree for the exchange to be meaningful -- the layout of the gradient arena is observed per rank."""
  if not (dist.is_available() and dist.is_initialized()) or not exchange_enabled(group):
    return
  t = torch.tensor([value, -value], dtype=torch.int64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  if int(t[0]) != -int(t[1]):
    raise RuntimeError(f'carla_garage_amd: the ranks disagree on {what}: the gradient exchange would mix different parameters')


def broadcast_state(flat_param, buffers, group=None, src=0):
  """Rank ``src``'s parameters and buffers everywhere (what the DDP constructor does, train.py:516)."""
  if not exchange_enabled(group):
    return
  dist.broadcast(flat_param, src, group=group)
  for b in buffers:
    dist.broadcast(b, src, group=group)


def _reduce_op(avg, group=None):
  """(op, post-scale): RCCL averages inside the collective (ReduceOp.AVG); gloo (the CPU tests) sums and the caller scales."""
  if not avg:
    return dist.ReduceOp.SUM, None
  if dist.get_backend(group) == 'nccl':
    return dist.ReduceOp.AVG, None
  return dist.ReduceOp.SUM, 1.0 / world_size(group)


def all_reduce_gradients(flat_grad, group=None, chunk_elems=None, avg=False):
  """SUM all-reduce of the flat gradient arena (the average is folded into the optimizer's grad_scale = 1/world), or with ``avg`` the
  mean (the drop-in path: torch optimizers expect averaged ``.grad`` as DistributedDataParallel leaves them, train.py:516-520).
  ``chunk_elems`` splits the arena into fewer, larger collectives than DDP's 25 MB buckets (default: one)."""
  w = world_size(group)
  if not exchange_enabled(group):
    return 1.0
  op, post = _reduce_op(avg, group)
  if chunk_elems is None or chunk_elems >= flat_grad.numel():
    dist.all_reduce(flat_grad, op=op, group=group)
  else:
    works = [dist.all_reduce(flat_grad[o:o + chunk_elems], op=op, group=group, async_op=True)
             for o in range(0, flat_grad.numel(), chunk_elems)]
    for wk in works:
      wk.wait()
  if post is not None:
    flat_grad.mul_(post)
  return 1.0 / w


class _ScaledWork:
  """Work handle of an averaged all-reduce on a backend without ReduceOp.AVG: wait, then scale."""

  def __init__(self, work, tensor, scale):
    self.work, self.tensor, self.scale = work, tensor, scale

  def wait(self):
    self.work.wait()
    self.tensor.mul_(self.scale)


def all_reduce_async(tensor, group=None, avg=False):
  """SUM (or mean) all-reduce of a slice of the gradient arena, returned as a work handle (``.wait()`` orders the current stream after
  it).  Used to send the part of the gradients that is finished early while the rest of backward still runs."""
  if not exchange_enabled(group) or tensor.numel() == 0:
    return None
  op, post = _reduce_op(avg, group)
  work = dist.all_reduce(tensor, op=op, group=group, async_op=True)
  return work if post is None else _ScaledWork(work, tensor, post)


def max_over_ranks(value, device, group=None):
  t = torch.tensor([float(value)], device=device, dtype=torch.float64)
  if world_size(group) > 1:
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  return float(t.item())


def rank_seed(base, rank):
  """Each rank draws a disjoint synthetic shard (DistributedSampler's role, train.py:544-553)."""
  return int(base) + int(rank)
