"""world_size-2 test of the data-parallel exchange on CPU (gloo): parameter broadcast, flat-gradient all-reduce with the
1/world average folded into the update, identical replicas afterwards, disjoint per-rank data seeds."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from carla_garage_amd import dist as tdist


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def _adamw_amsgrad_ref(p, g, m, v, vmax, lr, b1, b2, eps, wd, step, grad_scale):
  """Same arithmetic as csrc/misc_kernels.hip::adamw_amsgrad_kernel (torch.optim.AdamW(amsgrad=True) semantics)."""
  g = g * grad_scale
  p = p * (1 - lr * wd)
  m = b1 * m + (1 - b1) * g
  v = b2 * v + (1 - b2) * g * g
  vmax = torch.maximum(vmax, v)
  p = p - (lr / (1 - b1**step)) * m / (vmax.sqrt() / (1 - b2**step)**0.5 + eps)
  return p, m, v, vmax


def _worker(rank, world, port, out):
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  r, lr_, w = tdist.init_from_env('gloo')
  assert (r, w) == (rank, world) and tdist.world_size() == world
  n = 10007
  torch.manual_seed(100 + rank)  # different initial replicas on purpose
  flat_param = torch.randn(n)
  buf = torch.randn(5)
  tdist.broadcast_state(flat_param, [buf])
  m, v, vmax = torch.zeros(n), torch.zeros(n), torch.zeros(n)
  grads = []
  for step in range(1, 4):
    g = torch.Generator().manual_seed(tdist.rank_seed(1234, rank) * 10 + step)
    flat_grad = torch.randn(n, generator=g)
    grads.append(flat_grad.clone())
    if step == 3:  # the trainer's overlapped exchange: tail of the arena early (async), head afterwards, then wait
      early = tdist.all_reduce_async(flat_grad[6000:])
      assert early is not None
      scale = tdist.all_reduce_gradients(flat_grad[:6000])
      early.wait()
    else:
      scale = tdist.all_reduce_gradients(flat_grad, chunk_elems=4096 if step == 2 else None)
    assert scale == 1.0 / world
    flat_param, m, v, vmax = _adamw_amsgrad_ref(flat_param, flat_grad, m, v, vmax, 3e-4, 0.9, 0.999, 1e-8, 0.01, step, scale)
  tmax = tdist.max_over_ranks(1.0 + rank, torch.device('cpu'))
  torch.save({'param': flat_param, 'buf': buf, 'grads': grads, 'tmax': tmax}, os.path.join(out, f'rank{rank}.pt'))
  dist.destroy_process_group()


def test_two_rank_data_parallel_exchange(tmp_path):
  world, port = 2, _free_port()
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  r0 = torch.load(tmp_path / 'rank0.pt')
  r1 = torch.load(tmp_path / 'rank1.pt')
  assert torch.equal(r0['param'], r1['param'])  # replicas stay bit-identical
  assert torch.equal(r0['buf'], r1['buf'])
  assert r0['tmax'] == r1['tmax'] == 2.0
  assert not torch.equal(r0['grads'][0], r1['grads'][0])  # disjoint shards
  # single-process reference: average of the two ranks' gradients
  torch.manual_seed(100)
  p = torch.randn(10007)
  m, v, vmax = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
  for step in range(1, 4):
    g = r0['grads'][step - 1] + r1['grads'][step - 1]
    p, m, v, vmax = _adamw_amsgrad_ref(p, g, m, v, vmax, 3e-4, 0.9, 0.999, 1e-8, 0.01, step, 0.5)
  np.testing.assert_allclose(r0['param'].numpy(), p.numpy(), rtol=1e-6, atol=1e-7)


def test_single_process_is_a_no_op():
  t = torch.ones(8)
  assert tdist.all_reduce_gradients(t) == 1.0 and tdist.world_size() == 1
  tdist.broadcast_state(t, [])
  assert tdist.max_over_ranks(3.5, torch.device('cpu')) == 3.5
  assert tdist.all_reduce_async(t) is None


def _trainer_schedule_worker(rank, world, port, out):
  """Drives Trainer.train_step ITSELF (trainer.py: _step_part1 -> reduce_early -> _step_part2 -> finish_step) with the GPU parts stubbed: the
  backward segments only write this rank's gradients into the two slices of the arena.  What is tested is the bucket schedule: the early
  slice [early_offset:] is all-reduced asynchronously BETWEEN the segments, the head [:early_offset] after the second one, the optimizer runs
  after both have completed and sees the SUM over the ranks (the 1/world average is the optimizer's grad_scale)."""
  import types
  from carla_garage_amd.trainer import Trainer
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  tdist.init_from_env('gloo')
  n, off = 5003, 1801
  events = []
  tr = Trainer.__new__(Trainer)
  tr.model = types.SimpleNamespace(train=lambda: None)
  tr.pg, tr.world, tr.step_count, tr.exchange, tr._lazy_state, tr.early_opt_in_step, tr.no_decay_bits = None, world, 0, True, False, False, None
  tr.eng = types.SimpleNamespace(flat_grad=torch.zeros(n), early_offset=off, invalidate=lambda: None)
  mine = torch.arange(n, dtype=torch.float32) * (rank + 1)

  def part1(batch, split=True):
    assert split
    tr.eng.flat_grad[off:] = mine[off:]           # the heads / stage-4 gradients are final after the first backward segment ...
    events.append('segment1')
    return torch.zeros(3)

  def part2():
    assert 'early_issued' in events                # ... and are already travelling when the second segment starts
    tr.eng.flat_grad[:off] = mine[:off]
    events.append('segment2')

  def optimizer(step, grad_scale=None, upto=None, lo=0):
    # finish_step: the early slice [off:] is updated while the late slice travels, then the late slice [:off]
    assert (lo, upto) in ((off, None), (0, off))
    events.append('optimizer_early' if lo else 'optimizer_late')
    if lo:
      tr.seen = torch.zeros(n)
      tr.seen[lo:] = tr.eng.flat_grad[lo:]     # what this launch consumed: must already be the sum over the ranks
    else:
      tr.seen[:upto] = tr.eng.flat_grad[:upto]
    tr.scale = 1.0 / tr.world if grad_scale is None else grad_scale

  tr._step_part1, tr._step_part2, tr._optimizer = part1, part2, optimizer
  real_async, real_sync = tdist.all_reduce_async, tdist.all_reduce_gradients

  class Handle:

    def __init__(self, work, name='early_waited'):
      self.work, self.name = work, name

    def wait(self):
      events.append(self.name)
      self.work.wait()

  def rec_async(t, group=None, avg=False):
    if t.data_ptr() == tr.eng.flat_grad.data_ptr():
      assert t.numel() == off                                                             # exactly the head slice (late-finishing gradients)
      events.append('late_issued')
      return Handle(real_async(t, group, avg=avg), 'late_waited')
    assert t.data_ptr() == tr.eng.flat_grad[off:].data_ptr() and t.numel() == n - off   # exactly the tail slice of the arena
    events.append('early_issued')
    return Handle(real_async(t, group, avg=avg))

  def rec_sync(t, group=None, chunk_elems=None, avg=False):
    assert t.data_ptr() == tr.eng.flat_grad.data_ptr() and t.numel() == off               # exactly the head slice
    events.append('head_reduced')
    return real_sync(t, group, chunk_elems, avg=avg)

  tdist.all_reduce_async, tdist.all_reduce_gradients = rec_async, rec_sync
  assert tr.overlap_enabled()
  tr.train_step({})
  # the averaged variant the drop-in path uses (dropin.py): gloo has no ReduceOp.AVG -> SUM + scale after the wait
  g = mine.clone()
  h = real_async(g[off:], None, avg=True)
  real_sync(g[:off], None, avg=True)
  h.wait()
  torch.save({'events': events, 'seen': tr.seen, 'scale': tr.scale, 'avg': g}, os.path.join(out, f'sched{rank}.pt'))
  dist.destroy_process_group()


def test_trainer_bucket_schedule_two_ranks(tmp_path):
  world, port = 2, _free_port()
  mp.spawn(_trainer_schedule_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  want = torch.arange(5003, dtype=torch.float32) * 3.0  # rank 0 contributes 1x, rank 1 contributes 2x
  for r in range(world):
    d = torch.load(tmp_path / f'sched{r}.pt')
    # the early slice travels behind the second backward segment, the late slice behind the optimizer launch of the early slice
    assert d['events'] == ['segment1', 'early_issued', 'segment2', 'late_issued', 'early_waited', 'optimizer_early', 'late_waited', 'optimizer_late'], d['events']
    assert torch.equal(d['seen'], want) and d['scale'] == 0.5
    assert torch.equal(d['avg'], want / 2)
