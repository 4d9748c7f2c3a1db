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
  # ranks that disagree on something the exchange relies on (the arena layout is observed per rank) must ALL fail loudly, agreeing ranks pass
  tdist.assert_same_on_every_rank(123456789, 'a value every rank computes alike', torch.device('cpu'))
  try:
    tdist.assert_same_on_every_rank(1000 + rank, 'the layout of the gradient arena', torch.device('cpu'))
    raised = False
  except RuntimeError as e:
    raised = 'disagree on the layout' in str(e)
  assert raised
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
  """Drives Trainer.train_step ITSELF (trainer.py: _step_body -> finish_step -> GradBuckets.exchange) with the GPU parts stubbed: the step body
  only writes this rank's gradients into the arena.  What is tested is the bucket schedule: ONE step body (no second segment), then one
  asynchronous all-reduce per bucket of the arena in completion order, the optimizer launched on a bucket right after THAT bucket's
  collective has completed (never before), every bucket exactly once, and the optimizer sees the SUM over the ranks (the 1/world average is
  its grad_scale)."""
  import types
  from carla_garage_amd.buckets import GradBuckets
  from carla_garage_amd.trainer import Trainer
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  tdist.init_from_env('gloo')
  offsets = [0, 1792, 3072, 4224, 5003]  # four buckets (bucket starts are multiples of 128 elements, engine.BUCKET_ALIGN)
  n = offsets[-1]
  events = []
  tr = Trainer.__new__(Trainer)
  tr.model = types.SimpleNamespace(train=lambda: None, __dict__={})
  tr.pg, tr.world, tr.step_count, tr.exchange, tr._lazy_state, tr.no_decay_bits = None, world, 0, True, False, None
  tr.layout_final, tr.eager_steps_in_layout = True, 0
  buckets = GradBuckets()
  buckets.configure(offsets, 'cpu', observed=True)
  tr.eng = types.SimpleNamespace(flat_grad=torch.zeros(n), invalidate=lambda: None, buckets=buckets, observed_buckets=None)
  mine = torch.arange(n, dtype=torch.float32) * (rank + 1)

  def body(batch):
    tr.eng.flat_grad[:] = mine
    events.append('step_body')
    tr.program = ((0, 1, 2), (0, 1, 2))  # three buckets raise an early completion signal, the last one is complete at the end of the pass
    return torch.zeros(3)

  tr.seen = torch.zeros(n)

  def optimizer(step, grad_scale=None, upto=None, lo=0):
    b = offsets.index(lo)
    assert upto == offsets[b + 1]
    assert f'waited{b}' in events, (b, events)       # the optimizer never touches a bucket whose collective it has not waited for
    events.append(f'optimizer{b}')
    tr.seen[lo:upto] = tr.eng.flat_grad[lo:upto]     # what this launch consumed: must already be the sum over the ranks
    tr.scale = 1.0 / tr.world if grad_scale is None else grad_scale

  tr._step_body, tr._optimizer = body, optimizer
  real_async = tdist.all_reduce_async

  class Handle:

    def __init__(self, work, b):
      self.work, self.b = work, b

    def wait(self):
      events.append(f'waited{self.b}')
      self.work.wait()

  def rec_async(t, group=None, avg=False):
    b = [i for i in range(len(offsets) - 1) if t.data_ptr() == tr.eng.flat_grad[offsets[i]:].data_ptr()]
    assert len(b) == 1 and t.numel() == offsets[b[0] + 1] - offsets[b[0]]   # exactly one bucket of the arena
    events.append(f'issued{b[0]}')
    return Handle(real_async(t, group, avg=avg), b[0])

  tdist.all_reduce_async = rec_async
  assert tr.exchange_enabled()
  tr.train_step({})
  assert buckets.serial == 1  # the pass got serial 1 (GradBuckets.begin_issue): its completion signals carry that number, the waits look for it
  # the averaged variant the drop-in path uses (dropin.py): gloo has no ReduceOp.AVG -> SUM + scale after the wait
  tdist.all_reduce_async = real_async
  g = mine.clone()
  buckets.begin_issue()
  for w in buckets.exchange(g, tr.program, None, avg=True):
    w.wait()
  torch.save({'events': events, 'seen': tr.seen, 'scale': tr.scale, 'avg': g}, os.path.join(out, f'sched{rank}.pt'))
  dist.destroy_process_group()


def test_trainer_bucket_schedule_two_ranks(tmp_path):
  world, port = 2, _free_port()
  mp.spawn(_trainer_schedule_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  want = torch.arange(5003, dtype=torch.float32) * 3.0  # rank 0 contributes 1x, rank 1 contributes 2x
  for r in range(world):
    d = torch.load(tmp_path / f'sched{r}.pt')
    # one step body; the four all-reduces are issued back to back (each waits for its own completion signal on the device, not on the host);
    # then bucket after bucket: wait for its collective, update it -- the optimizer of bucket b overlaps the collectives of b + 1 ..
    assert d['events'] == ['step_body', 'issued0', 'issued1', 'issued2', 'issued3', 'waited0', 'optimizer0', 'waited1', 'optimizer1', 'waited2',
                           'optimizer2', 'waited3', 'optimizer3'], d['events']
    assert torch.equal(d['seen'], want) and d['scale'] == 0.5
    assert torch.equal(d['avg'], want / 2)
