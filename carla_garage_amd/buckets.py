"""Gradient buckets of the data-parallel exchange (team_code/train.py:516-520: DistributedDataParallel all-reduces ~20 buckets of 25 MB in
reverse order of the forward pass while backward still runs).

Here the flat gradient arena is laid out in the order in which the backward pass COMPLETES gradients (engine.arena_layout: one bucket per
batch of the weight-gradient lane, observed on a real pass), and a completion signal is raised behind the kernels that finish a bucket.
The exchange waits for nothing else: ONE captured graph, no second segment, no join of the lanes in the middle of backward.

The signal has to cross from a node inside a captured hipGraph to a stream outside of it.  ROCm 7.x has no host-side primitive for that
(tools/graph_external_event_test.py: PyTorch refuses external events on ROCm, hipEventRecordWithFlags(..., hipEventRecordExternal) fails
inside a capture, hipMallocSignalMemory / hipStreamWaitValue64 are not available on these boxes), so it is a counter in device memory:
``tfpp_signal_set`` is a one-thread kernel node behind the bucket's last kernels, ``tfpp_signal_wait`` a one-wave kernel on the collective's
stream that polls it (include/tfpp.h).  The same mechanism runs in eager steps, so the eager tests exercise what the graph replays.

Host bookkeeping (round 5: self-describing signals): a signal does not count passes, it CARRIES the serial number of the pass that
raised it.  ``begin_issue()`` -- called by whoever issues a pass, in front of the eager pass or of the graph replay, never inside a capture --
advances the host serial and writes it into a device word on the compute stream (``tfpp_set_u64``); the in-graph nodes raise their signal to
that word's value (``tfpp_signal_set``: max), and the exchange waits for "signal >= serial of THIS pass".  A pass nobody book-kept (a bare
``graph.replay()``, an eager pass that died half-way) re-raises an old serial, so a later wait can only be satisfied late -- a reported
time-out -- never early on gradients that are still being written.  A wait that times out is reported at the NEXT step on every rank: the
time-out word is MAX-all-reduced behind the collectives of EVERY exchange (unconditionally: whether a collective is posted must never depend
on a rank-local ``event.query()``, or the ranks' collective sequences diverge -- ADVICE r5) and copied to one slot of a small ring of pinned
host words; the host reads whichever slots have landed (``raise_if_timed_out``).

Correctness never depends on the observation being right: Engine.g() and the lane's flush hook cancel ("poison") the early signals of a
pass whose write order differs from the observed one; every bucket is then released by the signal raised at the END of the pass."""
import torch

from . import dist as tdist

import os

# a wait that sees nothing for this long gives up (and counts in `timeouts`): a stuck stream must not hang the GPU.  Far above any plausible
# step (a profiled / pre-empted / PMC-serialised step can take seconds: 2 s, the round-4 value, was within reach of those)
WAIT_TIMEOUT_MS = int(os.environ.get('TFPP_SIGNAL_TIMEOUT_MS', '20000'))
FLAG_RING = 4  # health flags in flight between the exchange stream and the host (one per exchange)


class GradBuckets:

  def __init__(self):
    self.offsets = [0, 0]
    self.device = None
    self.observed = False      # the layout comes from an observed pass (early signals are allowed)
    self.comm = None           # stream the collectives are issued from (RCCL's own stream waits for it, not for the compute stream)
    self.sig = None            # int64 [K + 1] device counters: one per bucket + the end-of-pass marker
    self.timeouts = None       # int32 [1]: waits that gave up
    self.serial = 0            # serial number of the pass issued last (begin_issue); the signals of a pass carry its serial
    self.serial_dev = None     # int64 [1] device word: the serial of the pass that is executing (written in front of it on the compute stream)
    self._flags = None         # ring of FLAG_RING (device word, pinned host word, event): time-out words on their way to the host
    self._flag_posted, self._flag_read, self._flag_seen = 0, 0, 0  # exchanges that posted a flag / flags the host has looked at / time-outs reported
    self._early = set()        # buckets with an early signal in the pass being recorded
    self.poisoned = None       # reason the early signals of this pass are not used
    self.stats = {'early_signals': 0, 'poisoned_passes': 0, 'exchanges': 0, 'waits': 0}

  @property
  def count(self):
    return len(self.offsets) - 1

  def configure(self, offsets, device, observed):
    self.offsets, self.device, self.observed = list(offsets), torch.device(device), bool(observed)
    if self.device.type == 'cuda':
      from . import ops
      # (a new layout gets new counters: graphs captured against the old ones are dead with the old arenas)
      self.sig = ops.zeros(self.count + 1, torch.int64, self.device)
      if self.timeouts is None:
        self.timeouts = ops.zeros(1, torch.int32, self.device)
      if self.serial_dev is None:
        self.serial_dev = ops.zeros(1, torch.int64, self.device)
        self.serial = 0
    self._early = set()

  def ranges(self):
    return [(self.offsets[b], self.offsets[b + 1]) for b in range(self.count)]

  # ------------------------------------------------------------------------------------------------ recording (backward pass)
  def begin_pass(self):
    self._early = set()
    self.poisoned = None

  def begin_issue(self):
    """A pass is about to be issued on the current stream (eager pass or graph replay): give it the next serial number and put that number
    where its signal nodes will read it.  NEVER inside a capture (the number would be frozen into the graph)."""
    self.serial += 1
    if self.serial_dev is not None:
      from . import ops
      assert not torch.cuda.is_current_stream_capturing(), 'GradBuckets.begin_issue() belongs in front of the capture / replay, not inside'
      ops.lib.tfpp_set_u64(self.serial_dev.data_ptr(), self.serial, torch.cuda.current_stream(self.device).cuda_stream)
    return self.serial

  def _raise(self, idx):
    from . import ops
    ops.lib.tfpp_signal_set(self.sig.data_ptr() + 8 * idx, self.serial_dev.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)

  def record(self, b):
    """Bucket b is complete behind everything issued so far on the CURRENT stream."""
    if self.poisoned is not None or not self.observed or not (0 <= b < self.count) or self.sig is None:
      return
    self._raise(b)
    self._early.add(b)
    self.stats['early_signals'] += 1

  def poison(self, why):
    """The write order of this pass differs from the observed one: release every bucket at the end of the pass only.  (Signals already
    raised in this pass stay part of its program -- the counters must advance the same way in every execution -- they are not waited on.)"""
    if self.poisoned is None:
      self.poisoned = why
      self.stats['poisoned_passes'] += 1

  def finish(self):
    """End of the pass, on the stream every lane has been joined into: raises the end-of-pass marker and returns the pass's program
    (raised early signals, usable early signals)."""
    raised = tuple(sorted(self._early))
    usable = raised if self.poisoned is None else ()
    if self.sig is not None:
      self._raise(self.count)
    return (raised, usable)

  def executed(self, program):
    """Kept for callers of the round-4 interface: the signals carry the serial of their pass (begin_issue), nothing is counted any more."""

  # ------------------------------------------------------------------------------------------------ exchange
  def exchange(self, flat_grad, program, group=None, avg=False):
    """One asynchronous all-reduce per bucket, in completion order, each ordered behind its bucket's signal only (the end-of-pass marker
    for buckets without a usable early signal).  Call after executed(program).  Returns the work handles (``wait()`` orders the current
    stream behind that bucket's collective), [] when no collective is issued."""
    if not tdist.exchange_enabled(group):
      return []
    self.raise_if_timed_out()  # (a wait of an EARLIER step that gave up: stop here, on every rank, before more gradients are averaged)
    self.stats['exchanges'] += 1
    _, usable = program
    works = []
    cuda = flat_grad.is_cuda and self.sig is not None
    if cuda and self.comm is None:
      from . import streams
      self.comm = streams.get(flat_grad.device, 'comm')
    from . import ops
    waited_end = False
    for b, (lo, hi) in enumerate(self.ranges()):
      if hi <= lo:
        works.append(None)
        continue
      if cuda:
        idx = b if b in usable else self.count
        if idx != self.count or not waited_end:  # (the stream is ordered: one wait on the end marker covers every later bucket)
          ops.lib.tfpp_signal_wait(self.sig.data_ptr() + 8 * idx, self.serial, WAIT_TIMEOUT_MS, self.timeouts.data_ptr(), self.comm.cuda_stream)
          self.stats['waits'] += 1
          waited_end |= idx == self.count
        with torch.cuda.stream(self.comm):
          works.append(tdist.all_reduce_async(flat_grad[lo:hi], group, avg=avg))
      else:
        works.append(tdist.all_reduce_async(flat_grad[lo:hi], group, avg=avg))
    if cuda:
      self._post_health_flag(group)
    return works

  # ------------------------------------------------------------------------------------------------ health of the exchange
  def _post_health_flag(self, group):
    """Behind this step's collectives on the exchange stream: MAX over the ranks of the time-out word, copied to pinned host memory.  Posted
    by EVERY exchange on every rank -- the collective sequence is a function of the step count alone -- into slot (exchange number mod
    FLAG_RING) of a ring; the host looks at the slots that have landed at the next step without waiting for the device (raise_if_timed_out)."""
    import torch.distributed as dist
    if self._flags is None:
      self._flags = [(torch.zeros(1, dtype=torch.int32, device=self.device), torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
                     for _ in range(FLAG_RING)]
    if self._flag_posted - self._flag_read >= FLAG_RING:
      # the slot about to be reused has not been read: the host is FLAG_RING exchanges ahead of the device.  Wait for that one copy (host-side
      # only: no collective is added or dropped) and read it
      self._read_flags(block=True, upto=self._flag_posted - FLAG_RING + 1)
    dev, host, ev = self._flags[self._flag_posted % FLAG_RING]
    with torch.cuda.stream(self.comm):
      dev.copy_(self.timeouts, non_blocking=True)
      if tdist.world_size(group) > 1:
        dist.all_reduce(dev, op=dist.ReduceOp.MAX, group=group)  # (every rank stops together)
      host.copy_(dev, non_blocking=True)
      ev.record(self.comm)
    self._flag_posted += 1

  def _read_flags(self, block, upto=None):
    """Look at the posted flags in order, oldest first, as far as they have landed (``block``: wait for them, all or up to number ``upto``);
    returns the largest time-out count seen."""
    worst = 0
    upto = self._flag_posted if upto is None else min(upto, self._flag_posted)
    while self._flag_read < upto:
      _, host, ev = self._flags[self._flag_read % FLAG_RING]
      if not ev.query():
        if not block:
          break
        ev.synchronize()
      worst = max(worst, int(host[0]))
      self._flag_read += 1
    return worst

  def raise_if_timed_out(self, block=False):
    """Raises when a completion-signal wait of an earlier step gave up: its all-reduce may have run on an incomplete bucket and the partial
    sums were averaged into every rank's gradients.  Cheap: reads the pinned host words the exchange stream has filled behind earlier
    collectives; it never waits for the device unless ``block`` (checkpoints: Trainer.check_exchange_health), and it never posts or skips a
    collective -- how often it is called per step does not matter."""
    if self._flags is None:
      return
    n = self._read_flags(block)
    if n > self._flag_seen:
      new, self._flag_seen = n - self._flag_seen, n
      raise RuntimeError(f'carla_garage_amd: {new} completion-signal wait(s) of the gradient exchange gave up after {WAIT_TIMEOUT_MS} ms (on this or another '
                         'rank): an all-reduce may have run on an incomplete gradient bucket; the gradients of that step are not trustworthy')

  def timed_out(self):
    """Number of signal waits that gave up since start-up (host synchronisation: tests / end of a run)."""
    return int(self.timeouts.item()) if self.timeouts is not None else 0
