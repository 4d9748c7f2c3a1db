"""Gradient buckets of the data-parallel exchange (team_code/train.py:516-520: DistributedDataParallel all-reduces ~20 buckets of 25 MB in
reverse order of the forward pass while backward still runs).

Here the flat gradient arena is laid out in the order in which the backward pass COMPLETES gradients (engine.arena_layout: one bucket per
batch of the weight-gradient lane, observed on a real pass), and a completion signal is raised behind the kernels that finish a bucket.
The exchange waits for nothing else: ONE captured graph, no second segment, no join of the lanes in the middle of backward.

The signal has to cross from a node inside a captured hipGraph to a stream outside of it.  ROCm 7.x has no host-side primitive for that
(tools/graph_external_event_test.py: PyTorch refuses external events on ROCm, hipEventRecordWithFlags(..., hipEventRecordExternal) fails
inside a capture, hipMallocSignalMemory / hipStreamWaitValue64 are not available on these boxes), so it is a counter in device memory:
``tfpp_signal_add`` is a one-thread kernel node behind the bucket's last kernels, ``tfpp_signal_wait`` a one-wave kernel on the collective's
stream that polls it (include/tfpp.h).  The same mechanism runs in eager steps, so the eager tests exercise what the graph replays.

Host bookkeeping: every executed pass (eager pass or graph replay) raises the same signals as the pass that was RECORDED -- its "program",
the tuple of buckets with an early signal, returned by finish() and kept by whoever owns the graph -- so `executed(program)` advances the
expected counter values once per executed pass, whether or not an exchange follows.

Correctness never depends on the observation being right: Engine.g() and the lane's flush hook cancel ("poison") the early signals of a
pass whose write order differs from the observed one; every bucket is then released by the signal raised at the END of the pass."""
import torch

from . import dist as tdist

WAIT_TIMEOUT_MS = 2000  # a wait that sees nothing for this long gives up (and counts in `timeouts`): a stuck stream must not hang the GPU


class GradBuckets:

  def __init__(self):
    self.offsets = [0, 0]
    self.device = None
    self.observed = False      # the layout comes from an observed pass (early signals are allowed)
    self.comm = None           # stream the collectives are issued from (RCCL's own stream waits for it, not for the compute stream)
    self.sig = None            # int64 [K + 1] device counters: one per bucket + the end-of-pass marker
    self.timeouts = None       # int32 [1]: waits that gave up
    self.expected = []         # host mirror of the counters: executed passes that raised each signal
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
    self.expected = [0] * (self.count + 1)
    self._early = set()

  def ranges(self):
    return [(self.offsets[b], self.offsets[b + 1]) for b in range(self.count)]

  # ------------------------------------------------------------------------------------------------ recording (backward pass)
  def begin_pass(self):
    self._early = set()
    self.poisoned = None

  def _raise(self, idx):
    from . import ops
    ops.lib.tfpp_signal_add(self.sig.data_ptr() + 8 * idx, torch.cuda.current_stream(self.device).cuda_stream)

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
    """A pass with this program has been issued (eager pass, or a replay of the graph it was captured into)."""
    raised, _ = program
    for b in raised:
      self.expected[b] += 1
    self.expected[self.count] += 1

  # ------------------------------------------------------------------------------------------------ exchange
  def exchange(self, flat_grad, program, group=None, avg=False):
    """One asynchronous all-reduce per bucket, in completion order, each ordered behind its bucket's signal only (the end-of-pass marker
    for buckets without a usable early signal).  Call after executed(program).  Returns the work handles (``wait()`` orders the current
    stream behind that bucket's collective), [] when no collective is issued."""
    if not tdist.exchange_enabled(group):
      return []
    self.stats['exchanges'] += 1
    _, usable = program
    works = []
    cuda = flat_grad.is_cuda and self.sig is not None
    if cuda and self.comm is None:
      self.comm = torch.cuda.Stream(flat_grad.device)
    from . import ops
    waited_end = False
    for b, (lo, hi) in enumerate(self.ranges()):
      if hi <= lo:
        works.append(None)
        continue
      if cuda:
        idx = b if b in usable else self.count
        if idx != self.count or not waited_end:  # (the stream is ordered: one wait on the end marker covers every later bucket)
          ops.lib.tfpp_signal_wait(self.sig.data_ptr() + 8 * idx, self.expected[idx], WAIT_TIMEOUT_MS, self.timeouts.data_ptr(), self.comm.cuda_stream)
          self.stats['waits'] += 1
          waited_end |= idx == self.count
        with torch.cuda.stream(self.comm):
          works.append(tdist.all_reduce_async(flat_grad[lo:hi], group, avg=avg))
      else:
        works.append(tdist.all_reduce_async(flat_grad[lo:hi], group, avg=avg))
    return works

  def timed_out(self):
    """Number of signal waits that gave up since start-up (host synchronisation: tests / end of a run)."""
    return int(self.timeouts.item()) if self.timeouts is not None else 0
