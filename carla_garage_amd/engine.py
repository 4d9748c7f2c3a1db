"""Execution engine of the MI355X TransFuser++ path: forward and hand-written backward over HIP kernels.

The reference expresses the model as ~1100 ATen calls recorded by torch.autograd
(team_code/model.py:279-392 and team_code/transfuser.py:139-257).  Here the same arithmetic is a static sequence of
launches of ``libtfpp_hip.so`` kernels (carla_garage_amd/ops.py) on NHWC activations; every forward primitive pushes
its own backward closure on a small tape, so the backward pass is again a static launch sequence with no ATen
kernels and can be captured into one hipGraph together with the optimizer step (carla_garage_amd/trainer.py).

Precision: the RegNet branches, fusion transformers and dense decoders run in ``dtype`` (fp32 or bf16 storage,
fp32 MFMA accumulation); the planning head (65-token decoder, GRU, MLPs: M <= 780 rows, launch-bound) always runs
in fp32.
"""
import contextlib
import math
import os

import torch

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_RELU, ACT_SIGMOID

F32 = torch.float32
EPS_F32 = float(torch.finfo(torch.float32).eps)


def _key(t):
  return (t.data_ptr(), t.numel())


# Round 6: train-mode BatchNorm without finalize / coefficient launches, conv1 / conv2 outputs of a bottleneck never normalised in memory
# (csrc/bn_rows_kernels.hip).  TFPP_BN_ROWS=0: the launch sequence of rounds 1-5 (A/B runs, comparison tests).
BN_ROWS = os.environ.get('TFPP_BN_ROWS', '1') != '0'


class BnView:
  """y = [relu](BatchNorm(raw)) that exists only as the raw convolution output plus the layer's statistics: its consumers (the 3x3 LDS-halo
  convolution and its weight gradient, the squeeze-excite passes) apply it while they load ``raw``; anything else calls
  Engine.materialize().  On the tape it stands for the virtual tensor y under the storage key of ``raw``.  ``final`` turns True once a
  consumer has run the finalize prologue (spec.scale / shift / save_mean / save_invstd are valid from then on)."""
  __slots__ = ('raw', 'spec', 'relu', 'rows', 'nrows', 'count', 'final')

  def __init__(self, raw, spec, relu, rows, nrows, count, final):
    self.raw, self.spec, self.relu, self.rows, self.nrows, self.count, self.final = raw, spec, relu, rows, nrows, count, final

  def data_ptr(self):
    return self.raw.data_ptr()

  def numel(self):
    return self.raw.numel()

  @property
  def shape(self):
    return self.raw.shape

  @property
  def dtype(self):
    return self.raw.dtype

  @property
  def device(self):
    return self.raw.device


DEBUG_POISON = os.environ.get('TFPP_DEBUG_POISON', '0') == '1'
# Engine.fast_weights_key: calls between two walks of the module tree (1: every call, +2 ms of host time per eval forward).  Only a tensor OBJECT
# replaced by attribute assignment on a sub-module (``conv.weight = nn.Parameter(...)``) can go unnoticed in between; in-place writes, load_state_dict,
# .to() / .cuda() / .float() and train() / eval() are seen at once.
FAST_KEY_RECHECK_EVERY = max(1, int(os.environ.get('TFPP_EVAL_GRAPH_RECHECK_EVERY', '64')))


def _release(tensors):
  """Drop the keep-alive references of tensors that crossed streams.  TFPP_DEBUG_POISON=1 (tools/stress_step.py): a tensor whose only
  remaining owner is this list is filled with NaN bytes first (on the current stream, i.e. after the joins that justify the release),
  so a consumer that was NOT ordered before the release computes on NaN instead of on silently recycled memory."""
  if DEBUG_POISON:
    import sys
    for i in range(len(tensors)):
      t = tensors[i]
      if t is not None and t.is_cuda and sys.getrefcount(t) <= 3 and t._base is None and t.is_contiguous():  # list slot + local + argument
        ops.lib.tfpp_fill_bytes(ops.ptr(t), 0xFF, t.numel() * t.element_size(), ops.stream())
      del t
  tensors.clear()


class Lanes:
  """HIP streams of the concurrent branches of the network.  Lane 0 is the caller's stream (image branch, fusion transformers,
  perspective decoders); lane 1 carries the LiDAR branch between the fusion points (a quarter of the pixels: small, latency-bound
  launches that overlap with the image branch) and afterwards the fp32 planning head.  Every tape node carries the lane it was
  recorded on, so backward mirrors the split.  Captured into the hipGraphs as parallel branches.  TFPP_BRANCH_STREAMS=0 keeps
  everything on one stream.  (A third lane for the BEV pyramid + CenterNet / BEV-semantic heads worked eagerly, but hipStreamEndCapture of
  ROCm 7.x segfaults on the resulting four-stream capture -- rounds 2 and 4 -- so those layers stay on lane 0.)"""

  def __init__(self):
    self.enabled = os.environ.get('TFPP_BRANCH_STREAMS', '1') != '0'
    self.main = None
    self.branches = {}    # lane -> torch.cuda.Stream (created on first use, kept for the life of the engine)
    self.active = set()   # lanes forked since the last join
    self.cur = 0
    self.held = []

  @property
  def branch(self):  # lane 1 (kept for callers that only know the two-lane layout)
    return self.branches.get(1)

  def hold(self, *tensors):
    """Keep tensors that cross lanes alive until the next pass begins: the caching allocator only orders reuse within
    the stream a block was allocated on, and the consumer runs on the other one."""
    if self.on():
      self.held.extend(tensors)

  def begin(self, device):
    """Call at the start of a forward / backward pass: lane 0 = the stream current right now."""
    self.cur = 0
    _release(self.held)
    self.held = []
    self.active = set()
    if not self.enabled or torch.device(device).type != 'cuda':
      self.main = None
      return
    self.main = torch.cuda.current_stream(device)

  def on(self):
    return self.main is not None

  def stream(self, k):
    if k == 0:
      return self.main
    st = self.branches.get(k)
    if st is None:
      from . import streams
      st = self.branches[k] = streams.get(self.main.device, 'lane', k)  # (process-wide: see streams.py)
    return st

  def touch(self, k, ev=None):
    """Lane k is about to receive work: the first time since the last join it has to wait for something of this pass (a stream that never
    waited on the capturing stream would not be part of a hipGraph capture) -- the event of the gradient it is about to consume when there is
    one (the lane then starts as soon as THAT is ready), else everything issued so far on lane 0."""
    if k != 0 and k not in self.active:
      if ev is not None:
        self.stream(k).wait_event(ev)
      else:
        self.stream(k).wait_stream(self.main)
      self.active.add(k)

  def streams(self):
    return [self.main] + [self.branches[k] for k in sorted(self.active)] if self.on() else []

  @contextlib.contextmanager
  def fork(self, k=1):
    """Run the body on lane k, ordered after everything issued so far on lane 0."""
    if not self.on() or k == 0:
      yield
      return
    st = self.stream(k)
    st.wait_stream(self.main)
    self.active.add(k)
    prev, self.cur = self.cur, k
    try:
      with torch.cuda.stream(st):
        yield
    finally:
      self.cur = prev

  def join(self):
    if self.on():
      for k in sorted(self.active):
        self.main.wait_stream(self.branches[k])


class Tape:
  """Reverse-mode tape: nodes are (outputs, inputs, backward_fn(*grad_outputs) -> grad_inputs, lane).

  Tensors are identified by (data_ptr, numel) so reshaped views of one buffer are the same node; the tape keeps every
  recorded tensor alive, so a key cannot be reused while it is pending.  With ``lanes`` a node's backward runs on the
  stream of the lane it was recorded on; a gradient that crosses lanes makes the consuming stream wait for the producing one."""

  current = None  # the tape whose backward() is running (closures use it to freeze tensors / register finalizers)

  def __init__(self, lanes=None):
    self.nodes = []
    self.lanes = lanes
    self.frozen = {}      # id(tensor) -> tensor: handed to another stream, must not be accumulated into in place
    self._graveyard = []
    self._grads = None    # pending gradients while backward() runs (take_pending)
    self.first = None     # (a, b): nodes recorded in [a, b) are walked FIRST in backward (Tape.hoist)
    self.on_finish = None  # callback: the whole backward pass has been issued
    self.finalizers = []  # run once at the end of backward (joins side streams)
    self.uses = {}        # key -> number of recorded nodes that consume the tensor (forward)
    self.contrib = {}     # key -> gradient contributions received so far (backward)
    self.on_accumulate = None  # callback(key): a gradient that was already handed out as "complete" received another addend

  def record(self, outs, ins, fn, lane=0):
    self.nodes.append((outs, ins, fn, lane))
    for t in ins:
      if t is not None:
        k = _key(t)
        self.uses[k] = self.uses.get(k, 0) + 1

  def is_last_contribution(self, t):
    """True while backward runs a node whose gradient for ``t`` is the last one ``t`` will receive: every other recorded consumer
    has already contributed.  The producer of that last addend then owns the COMPLETE gradient (with take_pending folded into its
    epilogue) and may compute reductions over it on the fly (fused BatchNorm-backward statistics)."""
    k = _key(t)
    return self.uses.get(k, 0) - self.contrib.get(k, 0) == 1

  def freeze(self, t):
    self.frozen[id(t)] = t

  def take_pending(self, t, like=None):
    """Remove and return the gradient already accumulated for tensor ``t`` (None if there is none, or if it must not be
    consumed: shared with another key, frozen for the weight-gradient lane, or of another dtype/size than ``like``).  The
    caller adds it inside its own kernel -- the residual operand of the data-gradient GEMM epilogue -- and returns the sum
    as the gradient of ``t``, which saves the separate accumulation pass over the tensor."""
    if self._grads is None:
      return None
    k = _key(t)
    cur = self._grads.get(k)
    if cur is None or self._refs.get(id(cur), 0) > 1 or id(cur) in self.frozen:
      return None
    if like is not None and (cur.dtype != like.dtype or cur.numel() != like.numel()):
      return None
    if self._multi and self._glane[k] != self._lane:
      self._wait(self._lane, k, self._glane[k])
      self.lanes.hold(cur)
    del self._grads[k]
    self._glane.pop(k, None)
    self._gevent.pop(k, None)
    self._refs[id(cur)] = self._refs.get(id(cur), 1) - 1
    return cur

  def hoist(self, a, b):
    """The nodes recorded in [a, b) form a block that depends only on the seeds (the planning head): walk it first in backward, whatever
    was recorded after it.  Inside a captured graph the order of capture decides which work a hardware queue picks up first; the block is
    ~2.7 ms of tiny launches on its own lane whose result the BEV pyramid and fusion stage 4 wait for, so it has to start with the pass."""
    self.first = (a, b)

  def relane(self, a, b, lane):
    """Backward runs the nodes recorded as a..b-1 on ``lane`` instead of the lane their forward ran on."""
    for i in range(a, min(b, len(self.nodes))):
      outs, ins, fn, _ = self.nodes[i]
      self.nodes[i] = (outs, ins, fn, lane)

  def backward(self, seeds):
    """seeds: list of (tensor, grad).  One pass over the recorded nodes in reverse order (the hoisted block first), the lanes joined at the end."""
    self._grads, self._refs, self._glane, self._lane = {}, {}, {}, 0
    self._gevent = {}  # key -> event recorded on the producing lane right after the gradient was written (see _wait)
    self.contrib = {}
    lanes = self.lanes
    if lanes is not None and seeds:
      lanes.begin(seeds[0][1].device)
    self._multi = lanes is not None and lanes.on()
    Tape.current = self
    for t, g in seeds:
      self._acc(t, g, 0)
    todo, self.nodes = self.nodes, []
    if self.first is not None and 0 <= self.first[0] < self.first[1] <= len(todo):
      a, b = self.first
      todo = todo[:a] + todo[b:] + todo[a:b]  # (_run walks the list backwards)
    self._run(todo)
    self._join()
    self._finish()

  def _sync(self, to_lane, from_lane):
    if self._multi and to_lane != from_lane:
      self.lanes.touch(to_lane)
      self.lanes.touch(from_lane)  # a producer lane idle since the last join joins this pass / capture first
      self.lanes.stream(to_lane).wait_stream(self.lanes.stream(from_lane))

  def _wait(self, to_lane, k, from_lane):
    """Lane ``to_lane`` is about to read (or add to) the pending gradient of key ``k`` written on ``from_lane``.  Fine-grained: wait for the EVENT
    recorded right after that gradient was written, not for the tail of the producing stream.  The tape is walked in reverse recording
    order, so by the time a LiDAR-lane node is issued the whole backward of the image stage recorded after it is already queued on lane 0:
    with stream-level waits the LiDAR stages, the planning head and the BEV heads ran strictly one after the other (round 3,
    tools/lane_timeline.py: lane 0 stalled 2.0 + 2.4 + 1.4 ms at the fusion points, lane 1 sat idle for 3.7 + 2.8 ms)."""
    if not self._multi or to_lane == from_lane:
      return
    ev = self._gevent.get(k)
    if ev is None:
      self._sync(to_lane, from_lane)
      return
    self.lanes.touch(to_lane, ev)
    self.lanes.stream(to_lane).wait_event(ev)

  def _mark_written(self, k, lane):
    if self._multi:
      ev = torch.cuda.Event()
      ev.record(self.lanes.stream(lane))
      self._gevent[k] = ev

  def _acc(self, t, g, lane):
    if t is None or g is None:
      return
    grads, refs, glane = self._grads, self._refs, self._glane
    k = _key(t)
    self.contrib[k] = self.contrib.get(k, 0) + 1
    cur = grads.get(k)
    if cur is not None and self.on_accumulate is not None:
      self.on_accumulate(k)
    if cur is None:
      grads[k] = g
      refs[id(g)] = refs.get(id(g), 0) + 1
      glane[k] = lane
      self._mark_written(k, lane)
      return
    if self._multi and glane[k] != lane:  # the pending gradient was last written on another lane
      self._wait(lane, k, glane[k])
      self.lanes.hold(cur, g)
    glane[k] = lane
    if refs.get(id(cur), 0) > 1 or id(cur) in self.frozen:
      # the stored gradient object is also pending under another key, or is being read on the side stream: do not mutate it
      refs[id(cur)] -= 1
      new = ops.add_dropout(cur, g if g.dtype == cur.dtype else ops.cast(g, cur.dtype))
      grads[k] = new
      refs[id(new)] = 1
    else:
      ops.axpy(g if g.dtype == cur.dtype else ops.cast(g, cur.dtype), cur, 1.0)
    self._mark_written(k, lane)

  def _run(self, nodes):
    grads, refs, glane, lanes, multi = self._grads, self._refs, self._glane, self.lanes, self._multi
    for pos in range(len(nodes) - 1, -1, -1):
      outs, ins, fn, lane = nodes[pos]
      if not multi:
        lane = 0
      gouts, src = [], []
      for o in outs:
        k = _key(o)
        g = grads.pop(k, None)
        if g is not None:
          refs[id(g)] = refs.get(id(g), 1) - 1
          src.append((k, glane.pop(k, 0)))
        gouts.append(g)
      if all(g is None for g in gouts):
        continue
      src_lanes = {sl for _, sl in src}
      ctx = torch.cuda.stream(lanes.stream(lane)) if multi and lane != 0 else contextlib.nullcontext()
      with ctx:
        self._lane = lane
        if multi:
          lanes.cur = lane
        for k, sl in src:
          self._wait(lane, k, sl)
          self._gevent.pop(k, None)
        if multi and lane != 0:
          lanes.touch(lane)  # (a lane whose first node of this pass consumes only its own gradients)
        if multi and src_lanes - {lane}:
          lanes.hold(*[g for g in gouts if g is not None])
        if ops.STAMPS['on']:
          ops.stamp(f'bwd lane{lane} {_fn_label(fn)}')
        if ops.NODE_HASH['on']:
          lab = _fn_label(fn)
          for j, g in enumerate(gouts):
            ops.node_hash(g, f'bwd lane{lane} {lab} consumes gout{j}')
        gins = fn(*gouts)
        if gins is None:
          gins = ()
        if not isinstance(gins, (tuple, list)):
          gins = (gins,)
        if ops.NODE_HASH['on']:
          for j, g in enumerate(gins):
            ops.node_hash(g, f'bwd lane{lane} {lab} produces gin{j}')
        for t, g in zip(ins, gins):
          self._acc(t, g, lane)
        if _KEEP_ALL:  # debugging aid: nothing the backward pass touched is freed (and so reused) before the pass ends
          self._graveyard.append((outs, ins, fn, gouts, gins))
    self._lane = 0
    if multi:
      lanes.cur = 0

  def _join(self):
    """End of the pass: flush + join the weight-gradient lane, join the encoder lanes."""
    ops.stamp('bwd lane0 MAIN CHAIN DONE (before the joins)')
    for fin in self.finalizers:
      fin()
    self.finalizers = []
    ops.stamp('bwd lane0 weight-gradient lane joined')
    if self._multi:
      self.lanes.join()
    ops.stamp('bwd lane0 all lanes joined')

  def _finish(self):
    if self.on_finish is not None:
      self.on_finish()
    if self._multi:
      _release(self.lanes.held)
      self.lanes.held = []
    self._grads = None
    self.frozen = {}
    self._graveyard = []
    Tape.current = None


def _fn_label(fn):
  """Readable name of a backward closure for the debugging tables: qualified name + the layer key / name it closed over."""
  lab = getattr(fn, '__qualname__', str(fn)).replace('.<locals>', '')
  try:
    for nme, cell in zip(fn.__code__.co_freevars, fn.__closure__ or ()):
      if nme in ('key', 'name', 'prefix') and isinstance(cell.cell_contents, str):
        lab += f'[{cell.cell_contents}]'
      elif nme == 's' and isinstance(cell.cell_contents, ConvSpec):
        lab += f'[{cell.cell_contents.name}]'
  except ValueError:  # empty cell
    pass
  return lab


# streams the weight-gradient lane alternates between, batch by batch; measured on the bs = 12 captured step (A/B/A/B, one box): 2: 24.38 / 24.43,
# 3: 24.03 / 24.02, 4 (every batch of the default three forks on its own stream): 23.76 / 23.99 ms/step
_SIDE_STREAMS = max(1, int(os.environ.get('TFPP_SIDE_STREAMS', '4')))


def _early_weights(name):
  """Layers that run before the first fusion point of the default TransFuser backbone (everything else is packed beside them)."""
  return any(name.startswith(p) for p in ('backbone.image_encoder.stem', 'backbone.image_encoder.s1.', 'backbone.lidar_encoder.stem',
                                          'backbone.lidar_encoder.s1.', 'backbone.lidar_channel_to_img.0'))
RELU_IN_DGRAD = os.environ.get('TFPP_RELU_IN_DGRAD', '1') != '0'  # ReLU backward of conv + bias + ReLU layers in the epilogue of the data gradient that completes their output gradient
_SKIP_SIDE_WORK = os.environ.get('TFPP_DEBUG_SKIP_SIDE_WORK', '0') == '1'
_SIDE_CHECK = os.environ.get('TFPP_DEBUG_SIDE_CHECK', '0') == '1'
_KEEP_ALL = os.environ.get('TFPP_DEBUG_KEEP_ALL', '0') == '1'
SIDE_OUT_LOG = []  # (layer, bias gradient from the lane, the same column sum recomputed after the join)
SIDE_CHECK_LOG = []  # (call site, sum / abs-sum at hand-over, sum / abs-sum at the join) device scalars of a captured step (TFPP_DEBUG_SIDE_CHECK=1)


class SideLane:
  """A second HIP stream for backward work that is off the critical path: weight gradients only feed the optimizer, so
  they trail the dY chain on their own stream and fill the CUs the small latency-bound kernels of the main chain leave
  idle.  Captured into the step's hipGraph as a parallel branch (fork = event wait, join at the end of backward).
  Tensors a side kernel reads are frozen on the tape (never accumulated into in place) and kept alive until the join, so
  the caching allocator cannot hand their memory to a main-stream kernel that might overtake the side stream.
  TFPP_SIDE_STREAM=0 runs everything on one stream."""

  def __init__(self):
    self.enabled = os.environ.get('TFPP_SIDE_STREAM', '1') != '0'
    # launches per fork (one event wait per batch); round 1: 8: 35.0, 16: 34.4, 32: 33.6, 64: 35.5 ms/step; end of round 2 (same box): 16: 30.9,
    # 32: 30.2-30.4, 64: 31.8, 128: 29.9-30.2, 256: 31.0; round 3 (same box, A/B/A/B): 32: 30.10 / 30.18, 128: 29.91 / 30.02.
    # Round 2 kept 32 because the replays of the captured step were not bit-reproducible at 96 / 128.  Round 3 found the cause
    # (tools/replay_bisect.py): not the lanes, but packed-FP32 VALU instructions in layernorm_bwd_kernel that return wrong row sums when the
    # wave shares its CU with the persistent MFMA weight-gradient kernel -- the library is now built without them (_lib.HIPCC_FLAGS) and
    # tools/stress_step.py passes at 96 and 128 (profiles/r03_stress_step_side_batch128.log).
    self.batch = int(os.environ.get('TFPP_SIDE_BATCH', '128'))
    self.tail_batch = int(os.environ.get('TFPP_SIDE_TAIL_BATCH', str(self.batch)))
    self.in_tail = False
    self.count = 0
    fa = os.environ.get('TFPP_SIDE_FLUSH_AT', '')
    self.flush_at = {int(v) for v in fa.split(',') if v} if fa else None
    self.forks = [float(v) for v in os.environ.get('TFPP_SIDE_FORKS', '0.46,0.80,0.95').split(',') if v and float(v) > 0]
    self.total_prev = 0  # closures of the previous pass (the eager warm-up in front of a capture counts them)
    self.last_flush_counts = []  # closure counts at the flushes of the previous pass
    self.stream = None
    self.streams, self.used, self.batches = [], set(), 0
    self.lanes = None  # Lanes of the engine: a batch may hold closures from both encoder-branch streams
    # gradient buckets (buckets.py): batches flushed so far in this pass, "a batch is running its closures right now", the closure count
    # at every flush of this pass, and the hook called on the batch's stream when a batch has been issued completely
    self.flush_seq, self.in_flush, self.flush_counts, self.on_batch_end = 0, False, [], None
    self.keep = []
    self.pending = []
    self.checks = []
    self.outs = []
    self.label = ''

  def run(self, tape, fn, *tensors):
    if _SKIP_SIDE_WORK:  # timing experiment only (gradients are wrong): how long is the step without any weight-gradient work?
      return
    if not self.enabled or tape is None or not tensors[0].is_cuda:
      fn()
      return
    if self.stream is None:
      from . import streams
      self.streams = [streams.get(tensors[0].device, 'side', j) for j in range(_SIDE_STREAMS)]  # (process-wide: see streams.py)
      self.stream = self.streams[0]
      self.used = set()
    if not self.keep:
      tape.finalizers.append(self.join)
    for t in tensors:
      tape.freeze(t)
      self.keep.append(t)
      if _SIDE_CHECK:  # debugging aid: nothing may write a tensor between its hand-over to this lane and the join
        import traceback
        where = f'{self.label} ' + ' <- '.join(f'{f.name}:{f.lineno}' for f in traceback.extract_stack(limit=5)[:-1][::-1])
        self.checks.append((t, t.double().sum(), t.double().abs().sum(), where))
    self.pending.append(fn)
    self.count += 1
    if self.flush_at is None and self.forks and self.total_prev:
      # fork points as fractions of the pass (the previous pass of this engine counted its closures): round 3 swept them on the bs = 12 step
      # (334 closures then; 374 in round 4, re-swept: 0.46 / 0.80 / 0.95 = closures 172 / 299 / 355, -0.17 ms against 0.43 / 0.77, profiles/r04_fork_sweep.txt):
      # TWO forks, in the middle of fusion transformer 3's backward (0.43) and when stage 3 of both encoders is done (0.77),
      # give 26.0 ms/step; any third fork costs ~2 ms, one fork ~2.5 ms, moving the second one 8 closures earlier 1.4 ms (tools/sweep_forks.sh)
      if self.count in {max(1, int(round(f * self.total_prev))) for f in self.forks}:
        self.flush()
      return
    if self.flush_at is not None:  # explicit fork points (closure counts since the start of the pass) instead of a fixed batch size
      if self.count in self.flush_at:
        self.flush()
      return
    # near the end of backward (stage 1 and the stems: the largest pixel counts, hence the longest weight-gradient kernels) whatever is still
    # queued when the main chain finishes is pure tail: fork in small batches there
    if len(self.pending) >= (self.tail_batch if self.in_tail else self.batch):
      self.flush()

  def flush(self, split=1):
    """Issue the queued closures as one batch.  ``split`` > 1 (the LAST batch of a pass: stage-1 / stem weight gradients, issued when the main
    chain is done and nothing else is left to run beside them) deals the closures round-robin onto that many streams of the lane: they are
    independent launches with a few workgroups each, and on one stream they run one after the other on an otherwise idle chip."""
    if self.pending:
      # successive batches alternate between the streams of the lane: the last batch of a pass then does not queue behind what is left of the
      # batch before it
      # all streams of the lane or one: dealing the batch onto SOME of them (2 or 3 of 4) runs eagerly but kills the process inside
      # hipStreamEndCapture of ROCm 7.2 when the step is captured (round 4; the same failure as a third branch lane, see Lanes)
      split = len(self.streams) if (split >= len(self.streams) and len(self.pending) >= len(self.streams)) else 1
      mine = [self.streams[(self.batches + j) % len(self.streams)] for j in range(split)]
      self.stream = mine[-1]
      self.batches += split
      seq = self.flush_seq
      self.flush_seq += 1
      self.flush_counts.append(self.count)
      for j, st in enumerate(mine):
        self.used.add(st)
        st.wait_stream(torch.cuda.current_stream())
        if self.lanes is not None:
          for ls in self.lanes.streams():
            st.wait_stream(ls)
        with torch.cuda.stream(st):
          if j == 0:
            ops.stamp(f'side lane9 batch of {len(self.pending)} begins')
          self.in_flush = True
          # the closures of a batch are independent layers: their pointwise weight gradients are collected and launched as grouped
          # grids (ops.wgrad_batch_end -> tfpp_conv_wgrad_batch) instead of two small launches per layer one after the other
          collecting = ops.wgrad_batch_begin()
          try:
            for fn in self.pending[j::split]:
              fn()
          finally:
            if collecting:
              ops.wgrad_batch_end()
            self.in_flush = False
      with torch.cuda.stream(self.stream):
        if self.on_batch_end is not None or ops.STAMPS['on']:
          # "every gradient of bucket `seq` is final" = this batch AND the earlier ones (other streams of the lane) are done; the waits sit
          # behind the batch's own kernels, so they delay the event, not the batch
          for st in self.used:
            if st is not self.stream:
              self.stream.wait_stream(st)
        ops.stamp('side lane9 batch ends')
        if self.on_batch_end is not None:
          self.on_batch_end(seq)
      self.pending = []

  def join(self):
    if self.keep:
      self.flush(split=len(self.streams))
      for st in self.used:
        torch.cuda.current_stream().wait_stream(st)
      self.used = set()
      self.batches = 0
      if _SIDE_CHECK and self.checks:
        if torch.cuda.is_current_stream_capturing():  # no host read inside a capture: leave device scalars for the caller to compare after a replay
          # bias gradients computed on this lane next to the same column sums recomputed on the caller's stream after the join
          SIDE_OUT_LOG.extend((k, g.clone(), dz.reshape(-1, dz.shape[-1]).float().sum(0)[:n].clone()) for k, dz, g, n in self.outs)
          SIDE_CHECK_LOG.extend((where, s0, a0, t.double().sum(), t.double().abs().sum()) for t, s0, a0, where in self.checks)
        else:
          bad = {}
          for t, s0, a0, where in self.checks:
            if float(t.double().sum()) != float(s0) or float(t.double().abs().sum()) != float(a0):
              bad[where] = bad.get(where, 0) + 1
          for where, n in bad.items():
            print(f'[TFPP_DEBUG_SIDE_CHECK] {n} tensor(s) handed to the weight-gradient lane were modified before the join: {where}', flush=True)
        self.checks = []
        self.outs = []
      _release(self.keep)
      self.keep = []
      self.total_prev, self.count = self.count, 0
      self.last_flush_counts, self.flush_counts, self.flush_seq = self.flush_counts, [], 0


# BatchNorm-backward sums (sum g, sum g*xhat) are produced by the kernel that completes the gradient where that kernel is the elementwise
# squeeze-excite backward (conv2 of every bottleneck); the data-gradient GEMMs can emit them as well (tfpp_conv_params.bns_*, tested at the
# op level) but the longer epilogues cost more on the chain than the reduction passes they replace (+0.75 ms/step, profiles/r04_switch_ab.txt),
# and a one-launch-per-sample squeeze-excite gate was 1.4 ms/step slower than the three small launches it replaced: neither is wired in.

EARLY_GRAD_PREFIXES = ('backbone.image_encoder.s4', 'backbone.lidar_encoder.s4', 'backbone.lidar_encoder.layers.layer3', 'backbone.lidar_encoder.norm', 'backbone.transformers.3', 'backbone.lidar_channel_to_img.3',
                       'backbone.img_channel_to_lidar.3', 'backbone.c5_conv', 'backbone.up_conv')
BUCKET_ALIGN = 128  # elements: a bucket starts on a whole word of the optimizer's no-decay bit mask (one bit per 4 elements)


def finishes_early(name):
  """Static guess used until the backward pass has been observed (Engine.end_backward): parameters outside the backbone (heads,
  decoders, planning head) and the stage-4 part of the backbone receive their gradients first -- about two thirds of the 481 MB."""
  return (not name.startswith('backbone.')) or name.startswith(EARLY_GRAD_PREFIXES)


def arena_layout(model):
  """Layout of the flat parameter / gradient arenas: ([(name, param, offset)], total elements, bucket offsets [o_0 = 0, ..., o_K = total]).

  Bucket b holds the parameters whose gradients are complete when the b-th batch of the weight-gradient lane has run (completion order
  of the backward pass), so the data-parallel exchange can all-reduce bucket b while the rest of backward still computes
  (team_code/train.py:516-520: what DistributedDataParallel does with ~20 reverse-order 25 MB buckets).  The assignment comes from an
  observed pass (``model._grad_buckets`` = {name: bucket}, set by Trainer.apply_observed_layout); before that, the static guess of
  finishes_early() gives two buckets.  Parameters are padded to 4 elements, buckets to BUCKET_ALIGN."""
  params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
  assign = model.__dict__.get('_grad_buckets')
  if assign is not None and all(n in assign for n, _ in params):
    key = lambda n: assign[n]
  else:
    key = lambda n: 0 if finishes_early(n) else 1
  nb = max((key(n) for n, _ in params), default=0) + 1
  out, offsets, off = [], [0], 0
  for b in range(nb):
    for n, p in params:
      if key(n) == b:
        out.append((n, p, off))
        off += ops.pad_to(p.numel(), 4)
    if b + 1 < nb:
      off = ops.pad_to(off, BUCKET_ALIGN)
    offsets.append(off)
  return out, off, offsets


def arena_order(model):
  """[(name, param)] in arena order (kept for tools: tools/replay_diff.py)."""
  return [(n, p) for n, p, _ in arena_layout(model)[0]]


class ConvSpec:
  """One convolution / linear layer: geometry, parameters, packed kernel images."""

  def __init__(self, name, weight, bias=None, bn=None, stride=1, pad=0, groups=1, cin_store=None, head=False):
    self.name = name
    self.weight, self.bias, self.bn = weight, bias, bn
    w4 = weight if weight.dim() == 4 else weight.view(weight.shape[0], -1, 1, 1)  # Linear [out, in]; Conv3d [out, in, kt, kh, kw] used as a GEMM
    self.cout, self.cin_g, self.k = w4.shape[0], w4.shape[1], w4.shape[2]
    self.stride, self.pad, self.groups = stride, pad, groups
    self.head = head  # fp32 planning-head layer
    # storage channel counts: inputs may be zero-padded (stem: 3 -> 8); outputs padded to a multiple of 8
    self.cin_store = cin_store if cin_store is not None else self.cin_g * groups
    self.n_store = self.cout if groups > 1 else ops.pad_to(self.cout, 8)
    self.wp = self.wt = self.bias_pad = None
    self.scale = self.shift = None  # folded BN (eval)
    self.row_map = None
    self.packed_for = None
    self.stat_rows = None   # fp32 [nrows][2 * n_store]: this layer's BatchNorm statistics rows, STORED by the conv epilogue (round 6)
    self.plan_cache = {}    # geometry -> (statistics rows of the kernel that runs it, can it normalise its source on load?)


class Engine:

  def __init__(self, model):
    self.m = model
    self.cfg = model.config
    self.dtype = F32
    self.tape = None
    self.training = False
    self.seed = 0x5EED
    self._seed_ctr = 0
    self.specs = {}
    self.grads = {}  # param name -> fp32 grad tensor (views of self.flat_grad)
    self.flat_grad = None
    self._consts = {}
    self._packed_key = None
    self._bn_of, self._bn_pre = {}, {}  # per forward: key(y) -> (spec, raw, relu) of conv+BN layers / key(y) -> fused backward sums
    self._relu_of, self._relu_done = {}, {}  # per forward: key(y) -> y of conv+bias+ReLU layers / key(y) -> the gradient that arrives already masked
    self._generation = 0
    self._plans, self._plan, self._plan_key = {}, None, None
    self._pack_stream, self._pack_pending = None, False
    self.lanes = Lanes()
    self.side = SideLane()
    self.side.lanes = self.lanes
    from .buckets import GradBuckets
    self.buckets = GradBuckets()
    self._glog, self._bucket_of, self._want_flushes, self.observed_buckets, self.observation_stable, self._prev_total = {}, None, None, None, False, 0
    self.bucket_program = ((), ())
    # SyncBatchNorm (train.py:511-512: nn.SyncBatchNorm.convert_sync_batchnorm(model) when config.sync_batch_norm = 1): BatchNorm statistics
    # over the batches of all ranks -- collectives inside the pass, so such a step is never captured into a hipGraph
    sbn = [mod for mod in model.modules() if isinstance(mod, torch.nn.SyncBatchNorm)]
    self.sync_bn = bool(sbn)
    self.sync_group = sbn[0].process_group if sbn else None
    self.sync_world = 1
    self._build_specs()

  # ------------------------------------------------------------------------------------------------ set-up
  def _spec(self, key, *a, **k):
    self.specs[key] = ConvSpec(key, *a, **k)
    return self.specs[key]

  def _build_specs(self):
    m = self.m
    bb = m.backbone
    self.aim = self.cfg.backbone == 'aim'  # team_code/aim.py: image branch only, no LiDAR branch, no fusion transformers
    self.bev = self.cfg.backbone == 'bev_encoder'  # team_code/bev_encoder.py: lift to BEV + second RegNet (bev.py)
    self.bev_runner = None
    if self.bev:
      from .bev import BevEncoderRunner
      self.bev_runner = BevEncoderRunner(self)
      self.bev_runner.build_specs()
    self.video = (not (self.aim or self.bev)) and getattr(bb, 'lidar_video', False)  # BASELINE config 5: Video-Swin LiDAR branch (swin.py)
    self.swin = None
    if self.video:
      from .swin import VideoSwin
      self.swin = VideoSwin(self)
      self.swin.build_specs()
    branches = [] if self.bev else [('image_encoder', bb.image_encoder)] + ([] if (self.aim or self.video) else [('lidar_encoder', bb.lidar_encoder)])
    for br, enc in branches:
      p = f'backbone.{br}'
      self._spec(f'{p}.stem', enc['stem'].conv.weight, bn=enc['stem'].bn, stride=2, pad=1, cin_store=8)
      for si in range(1, 5):
        for bname, blk in enc[f's{si}'].named_children():
          q = f'{p}.s{si}.{bname}'
          self._spec(q + '.conv1', blk.conv1.conv.weight, bn=blk.conv1.bn)
          self._spec(q + '.conv2', blk.conv2.conv.weight, bn=blk.conv2.bn, stride=blk.stride, pad=1,
                     groups=blk.conv2.conv.groups)
          self._spec(q + '.conv3', blk.conv3.conv.weight, bn=blk.conv3.bn)
          if blk.downsample is not None:
            self._spec(q + '.downsample', blk.downsample.conv.weight, bn=blk.downsample.bn, stride=blk.stride)
    for i in range(0 if (self.aim or self.bev) else 4):
      for nme in ('lidar_channel_to_img', 'img_channel_to_lidar'):
        conv = getattr(bb, nme)[i]
        self._spec(f'backbone.{nme}.{i}', conv.weight, conv.bias)
      g = bb.transformers[i]
      for l, blk in enumerate(g.blocks):
        q = f'backbone.transformers.{i}.blocks.{l}'
        self._spec(q + '.mlp.0', blk.mlp[0].weight, blk.mlp[0].bias)
        self._spec(q + '.mlp.2', blk.mlp[2].weight, blk.mlp[2].bias)
    if hasattr(bb, 'c5_conv'):
      self._spec('backbone.c5_conv', bb.c5_conv.weight, bb.c5_conv.bias)
      self._spec('backbone.up_conv5', bb.up_conv5.weight, bb.up_conv5.bias, pad=1)
      self._spec('backbone.up_conv4', bb.up_conv4.weight, bb.up_conv4.bias, pad=1)
    if self.cfg.detect_boxes:
      for n in m.head.BRANCHES:
        seq = getattr(m.head, n + '_head')
        self._spec(f'head.{n}_head.0', seq[0].weight, seq[0].bias, pad=1)
        self._spec(f'head.{n}_head.2', seq[2].weight, seq[2].bias)
    for dec_name in ('semantic_decoder', 'depth_decoder'):
      if hasattr(m, dec_name):
        dec = getattr(m, dec_name)
        for blk in ('deconv1', 'deconv2', 'deconv3'):
          for j in (0, 2):
            conv = getattr(dec, blk)[j]
            self._spec(f'{dec_name}.{blk}.{j}', conv.weight, conv.bias, pad=1)
    if self.cfg.use_bev_semantic:
      self._spec('bev_semantic_decoder.0', m.bev_semantic_decoder[0].weight, m.bev_semantic_decoder[0].bias, pad=1)
      self._spec('bev_semantic_decoder.2', m.bev_semantic_decoder[2].weight, m.bev_semantic_decoder[2].bias)
    self._spec('change_channel', m.change_channel.weight, m.change_channel.bias)
    # planning head (fp32): weights are used in place ([out,in] fp32 is already the kernel layout)
    h = dict(head=True)
    self._spec('extra_sensor_encoder.0', m.extra_sensor_encoder[0].weight, m.extra_sensor_encoder[0].bias, cin_store=8, **h)
    self._spec('extra_sensor_encoder.2', m.extra_sensor_encoder[2].weight, m.extra_sensor_encoder[2].bias, **h)
    self.tp_attention = bool(getattr(m, 'tp_attention', False))
    for l, layer in enumerate(m.join.layers):
      q = f'join.layers.{l}'
      if self.tp_attention:  # transfuser.py:404-443: separate key / query / value / proj linears per attention
        for an in ('self_attn', 'multihead_attn'):
          for ln_ in ('query', 'key', 'value', 'proj'):
            lin = getattr(getattr(layer, an), ln_)
            self._spec(f'{q}.{an}.{ln_}', lin.weight, lin.bias, **h)
      else:
        self._spec(q + '.self_attn.out_proj', layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias, **h)
        self._spec(q + '.multihead_attn.out_proj', layer.multihead_attn.out_proj.weight, layer.multihead_attn.out_proj.bias, **h)
      self._spec(q + '.linear1', layer.linear1.weight, layer.linear1.bias, **h)
      self._spec(q + '.linear2', layer.linear2.weight, layer.linear2.bias, **h)
    if self.tp_attention:
      self._spec('tp_encoder.0', m.tp_encoder[0].weight, m.tp_encoder[0].bias, cin_store=4, **h)
      self._spec('tp_encoder.2', m.tp_encoder[2].weight, m.tp_encoder[2].bias, **h)
    for dn in ('checkpoint_decoder', 'wp_decoder', 'wp_decoder_1'):
      if hasattr(m, dn):
        d = getattr(m, dn)
        self._spec(dn + '.gru.ih', d.gru.weight_ih_l0, d.gru.bias_ih_l0, **h)
        self._spec(dn + '.encoder', d.encoder.weight, d.encoder.bias, cin_store=4, **h)
    if hasattr(m, 'select_wps'):
      self._spec('select_wps', m.select_wps.weight, m.select_wps.bias, **h)
    if hasattr(m, 'target_speed_network'):
      self._spec('target_speed_network.0', m.target_speed_network[0].weight, m.target_speed_network[0].bias, **h)
      self._spec('target_speed_network.2', m.target_speed_network[2].weight, m.target_speed_network[2].bias, **h)

  def _const(self, key, fn):
    dev = self.device
    k = (key, str(dev))
    if k not in self._consts:
      self._consts[k] = fn().to(dev)
    return self._consts[k]

  @property
  def device(self):
    return self.m.change_channel.weight.device

  def next_seed(self):
    self._seed_ctr += 1
    return (self.seed * 1000003 + self._seed_ctr) & 0xFFFFFFFFFFFF

  # ------------------------------------------------------------------------------------------------ weights
  def _weights_key(self, dtype, need_t):
    return (dtype, need_t, self.training, self._generation, tuple((p.data_ptr(), p._version) for p in self.m.parameters()),
            tuple((b.data_ptr(), b._version) for b in self.m.buffers()) if not self.training else None)

  def fast_weights_key(self, dtype):
    """What _weights_key(dtype, False) watches in eval mode -- address and version of every parameter and buffer, the engine's generation -- over a CACHED
    list of the module's tensors: walking the module tree costs 2 ms of host time per call, reading 1332 addresses and versions 0.2 ms.  The list is
    rebuilt (and the key changes if a tensor OBJECT was replaced) whenever the module says its structure may have changed (LidarCenterNet bumps
    ``_structure_epoch`` in train() / eval(), load_state_dict() and _apply(): .to(), .cuda(), .float()) and on every FAST_KEY_RECHECK_EVERY-th call."""
    c = self.__dict__.get('_fast_key_cache')
    epoch = self.m.__dict__.get('_structure_epoch', 0)
    if c is None or c['epoch'] != epoch or c['calls'] % FAST_KEY_RECHECK_EVERY == FAST_KEY_RECHECK_EVERY - 1:
      tensors = list(self.m.parameters()) + list(self.m.buffers())
      ids = tuple(map(id, tensors))
      if c is None or ids != c['ids']:
        c = self.__dict__['_fast_key_cache'] = dict(tensors=tensors, ids=ids, serial=0 if c is None else c['serial'] + 1, calls=0)
      c['epoch'] = epoch
    c['calls'] += 1
    return (dtype, self._generation, c['serial'], tuple((t.data_ptr(), t._version) for t in c['tensors']))

  def invalidate(self):
    """Parameters / BatchNorm buffers were written through raw pointers (fused optimizer, BN running statistics, hipGraph
    replays): tensor._version cannot see that, so the trainer bumps the generation and the next prepare() repacks the weight
    images and re-folds the BatchNorm statistics."""
    self._generation += 1

  def prepare(self, dtype, training, need_grad):
    """(Re)pack weights for ``dtype`` when parameters changed; fold BN for eval."""
    self.dtype, self.training = dtype, training
    key = self._weights_key(dtype, need_grad)
    if key == self._packed_key:
      return
    self.repack(dtype, need_grad)
    self._packed_key = self._weights_key(dtype, need_grad)

  def repack(self, dtype, need_t, defer=False):
    """Refresh every kernel-layout weight image.  The first call for a (dtype, need_t) combination records all packing requests
    into a PackPlan with persistent destinations; later calls replay that plan as one launch.  Plans are kept per combination
    (train: forward + transposed images, eval: forward only), so alternating train / eval neither re-allocates the images nor
    frees buffers a captured hipGraph still writes into."""
    key = (dtype, need_t, str(self.device), tuple(p.data_ptr() for p in self.m.parameters()))
    ent = self._plans.get(key)
    if ent is None:
      plan = ops.PackPlan()
      ops.PACK_PLAN = plan
      try:
        self._repack_build(dtype, need_t)
      finally:
        ops.PACK_PLAN = None
      plan.finalize(self.device)
      ent = self._plans[key] = dict(plan=plan, specs={k: (s.wp, s.wt) for k, s in self.specs.items()},
                                    attn={k: {n: st.get(n) for n in self.ATTN_IMAGES} for k, st in getattr(self, '_attn', {}).items()})
    elif self._plan_key != key:  # switch the specs back to this plan's images
      for k, (wp, wt) in ent['specs'].items():
        self.specs[k].wp, self.specs[k].wt = wp, wt
      for k, imgs in ent['attn'].items():
        self._attn[k].update(imgs)
    self._plan, self._plan_key = ent['plan'], key
    if defer and torch.device(self.device).type == 'cuda':
      # the weight images of the stems and stage 1 now (the first ~1 ms of forward only needs those), the other ~99 % (0.4 ms of HBM streaming)
      # on their own stream beside that millisecond; Engine.forward waits for it at the first fusion point (_join_pack)
      ent['plan'].launch(0)
      if self._pack_stream is None:
        from . import streams
        self._pack_stream = streams.get(self.device, 'pack')
      cur = torch.cuda.current_stream(self.device)
      self._pack_stream.wait_stream(cur)
      with torch.cuda.stream(self._pack_stream):
        ent['plan'].launch(1)
        self._refresh_small()  # (padded biases of the heads / decoders: first read long after the join)
      self._pack_pending = True
    else:
      ent['plan'].launch()
      self._refresh_small()

  def _join_pack(self):
    if self._pack_pending:
      torch.cuda.current_stream(self.device).wait_stream(self._pack_stream)
      self._pack_pending = False

  ATTN_IMAGES = ('wqkv', 'bqkv', 'wproj', 'wqkv_t', 'wproj_t')

  def _refresh_small(self):
    """Padded biases and (eval) folded BatchNorm scale/shift: tiny per-layer launches."""
    for s in self.specs.values():
      if s.bias is not None and s.n_store != s.cout:
        ops.copy_rows(s.bias.detach(), s.bias_pad, 1, s.cout, 0, 0, 0, 0)
      elif s.bias is not None:
        s.bias_pad = s.bias.detach()
      if s.bn is not None and not (self.training and s.bn.training):
        ops.bn_fold(s.bn.weight.detach(), s.bn.bias.detach(), s.bn.running_mean, s.bn.running_var, s.scale, s.shift, s.bn.eps)

  def _repack_build(self, dtype, need_t):
    dev = self.device
    for s in self.specs.values():
      if ops.PACK_PLAN is not None:
        ops.PACK_PLAN.early = _early_weights(s.name)
      dt_ = F32 if s.head else dtype
      w = s.weight.detach()
      w4 = w if w.dim() == 4 else w.view(w.shape[0], -1, 1, 1)
      ks_pad = s.cin_store // s.groups
      plain = (dt_ == F32 and s.k == 1 and ks_pad == s.cin_g and s.n_store == s.cout and s.groups == 1)
      if plain:
        s.wp = w4.view(1, s.cout, s.cin_g)  # fp32 [out,in] is the kernel layout already
      else:
        s.wp = ops.pack_conv_weight(w4, dt_, G=s.groups, ks_pad=ks_pad, n_pad=s.n_store // s.groups)
      s.wt = ops.pack_conv_weight(w4, dt_, G=s.groups, n_pad=s.n_store // s.groups, transpose=True) if need_t else None
      if s.bias is not None and s.n_store != s.cout:
        if s.bias_pad is None or s.bias_pad.numel() != s.n_store or s.bias_pad.device != dev:
          s.bias_pad = ops.zeros(s.n_store, F32, dev)
      if s.n_store != s.cout and s.row_map is None:
        s.row_map = torch.tensor(list(range(s.cout)) + [-1] * (s.n_store - s.cout), dtype=torch.int32).to(dev)
      if s.bn is not None:
        if s.scale is None or s.scale.device != dev:
          s.scale = torch.empty(s.cout, device=dev, dtype=F32)
          s.shift = torch.empty(s.cout, device=dev, dtype=F32)
          s.save_mean = torch.empty(s.cout, device=dev, dtype=F32)
          s.save_invstd = torch.empty(s.cout, device=dev, dtype=F32)
          s.ws = torch.empty(2 * s.cout, device=dev, dtype=torch.float64)
    if ops.PACK_PLAN is not None:
      ops.PACK_PLAN.early = False
    # fusion-transformer QKV: fused, head-padded images
    for i, g in enumerate(getattr(self.m.backbone, 'transformers', [])):
      c, nh = g.n_embd, self.cfg.n_head
      d = c // nh
      dp = ops.pad_to(d, 8)

      def host_maps(nh=nh, d=d, dp=dp, c=c):
        r = torch.full((nh * dp,), -1, dtype=torch.int32)
        for hh in range(nh):
          r[hh * dp:hh * dp + d] = torch.arange(hh * d, (hh + 1) * d, dtype=torch.int32)
        iv = torch.full((c,), -1, dtype=torch.int32)  # parameter column -> packed column
        iv[r[r >= 0].long()] = torch.nonzero(r >= 0).flatten().to(torch.int32)
        return r, iv

      rmap = self._const(f'qkv_rmap{i}', lambda: host_maps()[0])
      inv = self._const(f'qkv_inv{i}', lambda: host_maps()[1])
      for l, blk in enumerate(g.blocks):
        st = self._attn_state(i, l)
        st.update(c=c, nh=nh, d=d, dp=dp, rmap=rmap, inv=inv)
        npk = 3 * nh * dp
        st['wqkv'] = torch.empty((npk, c), device=dev, dtype=dtype)
        st['bqkv'] = ops.zeros(npk, F32, dev)
        for j, lin in enumerate((blk.attn.query, blk.attn.key, blk.attn.value)):
          ops.pack2d(lin.weight.detach(), st['wqkv'], nh * dp, c, c, c, row_map=rmap, out_offset=j * nh * dp * c)
          ops.pack2d(lin.bias.detach(), st['bqkv'], 1, nh * dp, c, nh * dp, col_map=rmap, out_offset=j * nh * dp)
        st['wproj'] = torch.empty((c, nh * dp), device=dev, dtype=dtype)
        ops.pack2d(blk.attn.proj.weight.detach(), st['wproj'], c, nh * dp, c, nh * dp, col_map=rmap)
        if need_t:
          st['wqkv_t'] = torch.empty((c, npk), device=dev, dtype=dtype)
          for j, lin in enumerate((blk.attn.query, blk.attn.key, blk.attn.value)):
            ops.pack2d(lin.weight.detach(), st['wqkv_t'], c, nh * dp, c, npk, col_map=rmap, transpose_in=True,
                       out_offset=j * nh * dp)
          st['wproj_t'] = torch.empty((nh * dp, c), device=dev, dtype=dtype)
          ops.pack2d(blk.attn.proj.weight.detach(), st['wproj_t'], nh * dp, c, c, c, row_map=rmap, transpose_in=True)

  def _attn_state(self, i, l):
    if not hasattr(self, '_attn'):
      self._attn = {}
    return self._attn.setdefault((i, l), {})

  # ------------------------------------------------------------------------------------------------ gradients
  def alloc_grads(self, zero=True):
    """(Re)create the flat gradient arena when the set of trainable parameters or the bucket assignment changed; ``zero``: start the step
    from a zeroed arena (every gradient kernel accumulates).  The drop-in path zeroes (or keeps: gradient accumulation) the arena itself
    and passes zero=False."""
    sig = (tuple(id(p) for p in self.m.parameters() if p.requires_grad), id(self.m.__dict__.get('_grad_buckets')))
    if self.flat_grad is not None and sig == getattr(self, '_grad_sig', None) and self.flat_grad.device == self.device:
      if zero:
        if zero == 'beside_forward' and self._pack_pending:
          # 481 MB of memset that nothing reads before backward: on the weight-repack stream, beside the first layers of forward (joined at the
          # first fusion point, Engine._join_pack), instead of in front of the stem on lane 0
          with torch.cuda.stream(self._pack_stream):
            ops.zero_(self.flat_grad)
        else:
          ops.zero_(self.flat_grad)
        ops.clear_stats_rows(self.device)
      return
    self._grad_sig = sig
    layout, total, self.bucket_offsets = arena_layout(self.m)
    self.flat_grad = torch.empty(total, device=self.device, dtype=F32)  # (always a NEW tensor: holders of views compare identities)
    self.grads = {}
    for n, p, off in layout:
      self.grads[n] = self.flat_grad[off:off + p.numel()].view(p.shape)
    self._gid = {id(p): n for n, p, _ in layout}
    assign = self.m.__dict__.get('_grad_buckets')
    self._bucket_of = dict(assign) if assign is not None and all(n in assign for n in self.grads) else None  # None: static guess, nothing is known
    self.buckets.configure(self.bucket_offsets, self.device, observed=self._bucket_of is not None)
    ops.zero_(self.flat_grad)  # (a new arena always starts from zero)
    ops.clear_stats_rows(self.device)  # the fused BatchNorm statistics start every step from zeroed rows, whatever happened before

  def g(self, param):
    """Gradient slot of ``param`` (a view of the flat arena).  Every call is a write site of the backward pass: the batch of the
    weight-gradient lane that will have covered it is logged -- the arena is laid out by that order (arena_layout), and a write that
    comes LATER than the parameter's bucket says (another closure order than the observed one) cancels the early events of this pass."""
    n = self._gid.get(id(param))
    if n is None:
      return self._sink(param)
    side = self.side
    b = side.flush_seq - 1 if side.in_flush else side.flush_seq
    if b > self._glog.get(n, -1):
      self._glog[n] = b
      if self._bucket_of is not None and b > self._bucket_of[n]:
        self.buckets.poison(f'gradient of {n} written in batch {b}, its bucket is {self._bucket_of[n]}')
    return self.grads[n]

  def _sink(self, param):
    """Gradient slot of a parameter WITHOUT a slot in the arena (``requires_grad_(False)``: two-stage training, train.py:495-508, or any
    frozen sub-module between trainable ones).  Its backward closure still runs when a data gradient has to pass through the layer; what it
    computes for the parameter lands in one scratch buffer shared by all frozen parameters, which nobody reads.  (Sections of the network
    that are frozen AND fed by frozen sections only are not walked at all: Engine._drop_dead_nodes.)"""
    if param is None:
      return None
    buf = getattr(self, '_sink_buf', None)
    if buf is None or buf.numel() < param.numel() or buf.device != param.device:
      need = max([param.numel()] + [q.numel() for q in self.m.parameters() if not q.requires_grad])
      self._sink_buf = buf = torch.empty(need, device=param.device, dtype=F32)
    return buf[:param.numel()].view(param.shape)

  def _frozen(self, *modules):
    return all(not p.requires_grad for mod in modules if mod is not None for p in mod.parameters())

  def _drop_dead_nodes(self, start):
    """The nodes recorded since ``start`` belong to a section whose parameters are all frozen and whose inputs need no gradient either:
    autograd would not visit it (no leaf requires grad), neither does the tape."""
    tape = self.tape
    if tape is None:
      return
    del tape.nodes[start:]

  def begin_backward(self):
    """Start of a backward pass (Trainer / DropinStep call it before Tape.backward)."""
    self._glog = {}
    self.buckets.begin_pass()
    self._prev_total = self.side.total_prev
    self.side.last_flush_counts = []
    self.side.on_batch_end = self._batch_end
    # the closure counts at which the lane flushed in the observed pass: another flush pattern = other bucket contents
    self._want_flushes = self.m.__dict__.get('_grad_bucket_flushes')

  def _batch_end(self, seq):
    side = self.side
    if self._bucket_of is None:
      return
    want = self._want_flushes
    if want is None or seq >= len(want) or side.flush_counts[seq] != want[seq]:
      self.buckets.poison(f'batch {seq} of the weight-gradient lane was flushed at closure {side.flush_counts[seq]}, the observed pass flushed at {want}')
      return
    if seq < self.buckets.count - 1:  # (the last bucket is complete at the end of the pass: buckets.finish())
      self.buckets.record(seq)

  def end_backward(self):
    """End of a backward pass (after Tape.backward returned: every lane joined).  Returns the observation of this pass:
    ({name: bucket}, closure counts at the flushes) -- what Trainer.apply_observed_layout turns into the arena layout."""
    self.bucket_program = self.buckets.finish()  # (early signals raised, early signals usable): kept by whoever replays this pass
    self.side.on_batch_end = None
    flushes = list(self.side.last_flush_counts)
    obs = dict(self._glog)
    nb = max(len(flushes), 1)
    for n in self.grads:  # parameters nobody wrote (unused heads): with the last bucket
      obs[n] = min(obs.get(n, nb - 1), nb - 1)
    self.observed_buckets = (obs, flushes)
    # the lane forks at fractions of the PREVIOUS pass's closure count: the observation describes the next pass once that count has settled
    side = self.side
    self.observation_stable = (not side.enabled) or (not side.forks) or side.flush_at is not None or (side.total_prev > 0 and side.total_prev == self._prev_total)
    return self.observed_buckets

  # ------------------------------------------------------------------------------------------------ primitives
  def rec(self, outs, ins, fn):
    if self.tape is not None:
      self.tape.record(outs, ins, fn, self.lanes.cur)
      if ops.STAMPS['on']:
        ops.stamp(f'fwd lane{self.lanes.cur} {_fn_label(fn)}')
      if ops.NODE_HASH['on']:
        for j, o in enumerate(outs):
          ops.node_hash(o.raw if isinstance(o, BnView) else o, f'fwd lane{self.lanes.cur} {_fn_label(fn)} out{j}')

  # ---- round 6: BatchNorm(train) layers whose statistics live in per-layer rows until a consumer needs them --------------------------------
  def _bn_desc(self, v, finalize=True):
    """tfpp_bn_rows of a BnView for the next kernel that reads it: the first consumer gets the rows (finalize prologue: it also writes scale /
    shift / saved statistics and updates the running statistics), every later one reads scale / shift."""
    sp = v.spec
    if finalize and not v.final:
      v.final = True
      bn = sp.bn
      return ops.bn_rows(sp.n_store, sp.scale, sp.shift, partial=v.rows, nrows=v.nrows, count=v.count, gamma=bn.weight.detach(), beta=bn.bias.detach(),
                         rm=bn.running_mean, rv=bn.running_var, nbt=bn.num_batches_tracked, save_mean=sp.save_mean, save_invstd=sp.save_invstd,
                         momentum=bn.momentum, eps=bn.eps)
    assert v.final, 'a BatchNorm view is read before any consumer finalized its statistics'
    return ops.bn_rows(sp.n_store, sp.scale, sp.shift)

  def materialize(self, v):
    """The normalised tensor of a BnView in memory, for consumers that cannot apply it while loading (LDS-DMA GEMMs, strided 3x3 weight
    gradients, maps narrower than a halo tile)."""
    if not isinstance(v, BnView):
      return v
    y = ops.bn_apply_rows(v.raw, self._bn_desc(v), relu_pre=v.relu)
    if self.tape is not None:
      self.rec([y], [v], lambda dy: dy)  # (the producer's backward takes d(y) and rebuilds the ReLU mask from raw)
    return y

  def _conv_plan(self, s, xt, geo, want_grad):
    """(statistics rows of the kernel that runs this layer, can that kernel -- and the layer's weight-gradient kernel -- normalise the source
    while loading it?) -- cached per geometry."""
    k = (tuple(xt.shape), xt.dtype, want_grad)
    plan = s.plan_cache.get(k)
    if plan is None:
      nrows = ops.conv_gemm(xt, s.wp, None, stats_rows_query=True, **geo) if s.bn is not None else 0
      on_load = xt.dtype == torch.bfloat16 and ops.conv_gemm(xt, s.wp, None, in_bn_query=True, **geo)
      if on_load and want_grad:
        on_load = ops.conv_wgrad_x_bn_ok(xt, xt, s.scale if s.scale is not None else torch.empty(1, device=xt.device), B=geo['B'], Hs=geo['Hs'], Ws=geo['Ws'],
                                         Cs=geo['Cs'], Hd=geo['Hd'], Wd=geo['Wd'], Cd=geo['Cd'], R=geo['R'], S=geo['S'], stride=geo['stride'], pad=geo['pad'],
                                         G=geo['G'], ks_g=geo['ks_g'], n_g=geo['n_g'], c_real=s.cin_g)
      plan = s.plan_cache[k] = (nrows, bool(on_load))
    return plan

  def conv(self, x, key, act=ACT_NONE, res=None, x_grad=True, out_f32=False, lazy=False):
    """x: [B,H,W,Cstore] NHWC, or a BnView (the un-normalised output of the BatchNorm layer in front).  Returns [B,Ho,Wo,n_store]; with
    ``lazy`` a train-mode BatchNorm layer returns a BnView instead of writing the normalised tensor (the caller's next operator consumes it)."""
    s = self.specs[key]
    xv = x if isinstance(x, BnView) else None
    xt = xv.raw if xv is not None else x
    B, H, W, Cs = xt.shape
    k, st, pd, G = s.k, s.stride, s.pad, s.groups
    Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
    odt = F32 if (out_f32 or s.head) else xt.dtype
    geo = dict(B=B, Hs=H, Ws=W, Cs=Cs, Hd=Ho, Wd=Wo, Cd=s.n_store, R=k, S=k, stride=st, pad=pd, G=G, ks_g=Cs // G,
               n_g=s.n_store // G)
    bn_train = s.bn is not None and self.training and s.bn.training
    sync = self.sync_bn and self.sync_world > 1
    want_grad = self.tape is not None and s.weight.requires_grad
    if xv is not None and not self._conv_plan(s, xt, geo, want_grad)[1]:
      x, xv = self.materialize(xv), None
      xt = x
    in_kw = dict(in_bn=self._bn_desc(xv), in_relu=xv.relu) if xv is not None else {}
    rows_path = bn_train and BN_ROWS and not sync and act in (ACT_NONE, ACT_RELU) and odt == xt.dtype and s.n_store == s.cout
    view = None
    if s.bn is None:
      y = torch.empty((B, Ho, Wo, s.n_store), device=xt.device, dtype=odt)
      ops.conv_gemm(xt, s.wp, y, act=act, shift=s.bias_pad, res=res, **in_kw, **geo)
      raw = None
      if RELU_IN_DGRAD and self.tape is not None and act == ACT_RELU and odt == torch.bfloat16:
        self._relu_of[_key(y)] = y  # the data gradient that completes d(y) may apply this layer's ReLU backward in its epilogue
    elif not bn_train:
      if self.tape is not None:
        raise NotImplementedError('gradients through eval-mode BatchNorm are not implemented on the HIP path (round 1); '
                                  'call model.train() or run under torch.no_grad()')
      y = torch.empty((B, Ho, Wo, s.n_store), device=xt.device, dtype=odt)
      ops.conv_gemm(xt, s.wp, y, act=act, scale=s.scale, shift=s.shift, res=res, **in_kw, **geo)
      raw = None
    elif rows_path:
      # conv (statistics rows in the epilogue) and nothing else: the finalize step runs in the prologue of whoever reads the result first
      raw = torch.empty((B, Ho, Wo, s.n_store), device=xt.device, dtype=odt)
      nrows = self._conv_plan(s, xt, geo, want_grad)[0]
      if nrows <= ops.BN_ROWS_MAX:
        if s.stat_rows is None or s.stat_rows.numel() < nrows * 2 * s.n_store or s.stat_rows.device != xt.device:
          if xt.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('BatchNorm statistics rows must be allocated before hipGraph capture: run one eager warm-up step first')
          s.stat_rows = torch.empty(nrows * 2 * s.n_store, device=xt.device, dtype=F32)
        ops.conv_gemm(xt, s.wp, raw, stats_store=s.stat_rows, **in_kw, **geo)
        view = BnView(raw, s, act == ACT_RELU, s.stat_rows, nrows, B * Ho * Wo, False)
      else:  # large feature maps (stage 1, the stems): too many rows for a prologue -- the finalize launch of rounds 1-5
        nrows, acc = ops.conv_gemm(xt, s.wp, raw, stats_acc=True, **in_kw, **geo)
        ops.bn_finalize_partials(acc, nrows, s.bn.weight.detach(), s.bn.bias.detach(), s.bn.running_mean, s.bn.running_var,
                                 s.bn.num_batches_tracked, s.scale, s.shift, s.save_mean, s.save_invstd, B * Ho * Wo,
                                 s.bn.momentum, s.bn.eps)
        view = BnView(raw, s, act == ACT_RELU, None, 0, B * Ho * Wo, True)
      if lazy and res is None:
        y = view
      else:
        y = ops.bn_apply_rows(raw, self._bn_desc(view), res=res, relu_pre=(act == ACT_RELU and res is None), relu_post=(act == ACT_RELU and res is not None))
    else:
      raw = torch.empty((B, Ho, Wo, s.n_store), device=xt.device, dtype=odt)
      nrows, acc = ops.conv_gemm(xt, s.wp, raw, stats_acc=True, **in_kw, **geo)  # BN statistics fused into the epilogue
      if sync:  # train.py:511-512: statistics over the batches of all ranks (one all-reduce of 2C doubles per layer)
        ops.bn_sync_finalize(acc, nrows, s.bn.weight.detach(), s.bn.bias.detach(), s.bn.running_mean, s.bn.running_var, s.bn.num_batches_tracked,
                             s.scale, s.shift, s.save_mean, s.save_invstd, B * Ho * Wo, self.sync_world, self.sync_group, s.ws, s.bn.momentum, s.bn.eps)
      else:
        ops.bn_finalize_partials(acc, nrows, s.bn.weight.detach(), s.bn.bias.detach(), s.bn.running_mean, s.bn.running_var,
                                 s.bn.num_batches_tracked, s.scale, s.shift, s.save_mean, s.save_invstd, B * Ho * Wo,
                                 s.bn.momentum, s.bn.eps)
      y = ops.affine_act(raw, scale=s.scale, shift=s.shift, res=res, act=act)
      if self.tape is not None and act in (ACT_NONE, ACT_RELU):
        self._bn_of[_key(y)] = (s, raw, act == ACT_RELU)  # lets the producer of d(y) fuse this layer's BatchNorm-backward sums
    if self.tape is not None:
      xin = xv if xv is not None else x  # what the tape knows the input as
      wg_kw = dict(x_scale=xv.spec.scale, x_shift=xv.spec.shift, x_relu=xv.relu) if xv is not None else {}

      def bwd(dy):
        self.side.label = key
        self.side.in_tail = ('.s1.' in key or key.endswith('.stem'))
        if s.bn is None:
          masked = self._relu_done.pop(_key(y), None) is dy  # the kernel that wrote dy already zeroed it where y <= 0
          dz = ops.act_bwd(dy, y, act) if (act != ACT_NONE and not masked) else dy
          dres = dz if res is not None else None
          if s.bias is not None and s.bias.requires_grad:

            def bias_grad():  # parameter gradient only: weight-gradient lane
              if s.n_store == s.cout:
                ops.colsum(dz, self.g(s.bias), B * Ho * Wo, s.cout, s.n_store)
              else:
                tmp = ops.zeros(s.n_store, F32, xt.device)
                ops.colsum(dz, tmp, B * Ho * Wo, s.n_store, s.n_store)
                ops.copy_rows(tmp, self.g(s.bias), 1, s.cout, 0, 0, 0, 0, accumulate=True)

            self.side.run(Tape.current, bias_grad, dz)
            if _SIDE_CHECK:
              self.side.outs.append((key, dz, self.g(s.bias), s.cout))
          dconv = dz
        elif view is not None:
          # reduce (unless the kernel that completed dy already emitted the sums) -> apply with the coefficient step in its prologue.  The ReLU
          # mask comes from the forward output where one exists in memory with a residual folded in, otherwise it is rebuilt from raw
          pre = self._bn_pre.pop(_key(y), None)
          relu = act == ACT_RELU
          mask = ops.MASK_NONE if not relu else (ops.MASK_Y if res is not None else ops.MASK_RAW)
          yk = y if mask == ops.MASK_Y else None
          if pre is not None and pre[2] is dy:
            partial, nrows_b = pre[0], pre[1]
          else:
            partial, nrows_b = ops.bn_bwd_reduce_rows(dy, yk, raw, s.scale, s.shift, s.save_mean, s.save_invstd, mask)
          dconv, dres = ops.bn_bwd_apply_rows2(dy, yk, raw, s.scale, s.shift, s.bn.weight.detach(), s.save_mean, s.save_invstd, partial, nrows_b,
                                               self.g(s.bn.weight), self.g(s.bn.bias), mask, want_dres=res is not None)
        elif bn_train:
          pre = self._bn_pre.pop(_key(y), None)
          if sync:
            dconv, dres = ops.bn_bwd_sync(dy, y, raw, s.bn.weight.detach(), s.save_mean, s.save_invstd, self.g(s.bn.weight), self.g(s.bn.bias),
                                          relu_mask=(act == ACT_RELU), world=self.sync_world, group=self.sync_group, want_dres=res is not None)
          elif pre is not None and pre[2] is dy:  # the kernel that wrote dy already reduced sum g / sum g*xhat per tile (one pass saved)
            dconv, dres = ops.bn_bwd_rows(dy, y, raw, s.bn.weight.detach(), s.save_mean, s.save_invstd, pre[0], pre[1], self.g(s.bn.weight),
                                          self.g(s.bn.bias), relu_mask=(act == ACT_RELU), want_dres=res is not None)
          else:
            dconv, dres = ops.bn_bwd(dy, y, raw, s.bn.weight.detach(), s.save_mean, s.save_invstd, None, self.g(s.bn.weight),
                                     self.g(s.bn.bias), relu_mask=(act == ACT_RELU), want_dres=res is not None)
        gsrc = dconv if dconv.dtype == xt.dtype else ops.cast(dconv, xt.dtype)
        if s.weight.requires_grad:
          self.side.run(Tape.current, lambda: ops.conv_wgrad(
              gsrc, xt, self.g(s.weight), B=B, Hs=H, Ws=W, Cs=Cs, Hd=Ho, Wd=Wo, Cd=s.n_store, R=k, S=k, stride=st, pad=pd, G=G,
              ks_g=Cs // G, n_g=s.n_store // G, c_real=s.cin_g, row_map=s.row_map, **wg_kw), gsrc, xt)
        dx = None
        if x_grad:
          dx = torch.empty((B, H, W, Cs), device=xt.device, dtype=xt.dtype)
          # a gradient already pending for x (the other path of a residual / FPN fan-out) is added in the GEMM epilogue
          last = Tape.current.is_last_contribution(xin)
          pend = Tape.current.take_pending(xin, dx)
          dgeo = dict(B=B, Hs=Ho, Ws=Wo, Cs=s.n_store, Hd=H, Wd=W, Cd=Cs, R=k, S=k, stride=st, pad=pd, G=G, ks_g=s.n_store // G, n_g=Cs // G,
                      mode=1, res=pend)
          # x = relu(conv + bias) of a layer without BatchNorm and this is the last addend of d(x): that layer's ReLU backward runs here
          fwd_x = self._relu_of.get(_key(xin)) if xv is None else None
          if fwd_x is not None and last:  # (masking is idempotent: an addend that still arrived later would only bring the separate pass back)
            mk = ('relu_mask', tuple(xt.shape), pend is not None)
            ok = s.plan_cache.get(mk)
            if ok is None:
              ok = s.plan_cache[mk] = ops.conv_gemm(gsrc, s.wt, dx, relu_mask_query=True, **dgeo)
            if ok:
              dgeo['relu_mask'] = fwd_x
              self._relu_done[_key(xin)] = dx
          ops.conv_gemm(gsrc, s.wt, dx, **dgeo)
        return dx, dres

      self.rec([y], [xin, res], bwd)
    return y

  def linear(self, x, key, act=ACT_NONE, res=None, x_grad=True, out_f32=False):
    """x: [..., K] tokens -> [..., n_store]."""
    shp = x.shape
    rows = x.numel() // shp[-1]
    y = self.conv(x.view(rows, 1, 1, shp[-1]), key, act=act, res=None if res is None else res.view(rows, 1, 1, -1),
                  x_grad=x_grad, out_f32=out_f32)
    return y.view(*shp[:-1], y.shape[-1])

  def raw_linear(self, x, w, bias, wt, gw, gb, n, k, act=ACT_NONE, row_map=None, col_map=None, n_real=None, res=None):
    """Linear with explicitly supplied packed images (fusion QKV / proj, decoder in_proj slices).
    x: [rows, k];  w: [n, k];  wt: [k, n] (for the data gradient);  gw/gb: gradient destinations (callables);
    res: [rows, n] added in the GEMM epilogue (the residual stream of a transformer block when no dropout sits in between)."""
    rows = x.numel() // k
    y = torch.empty((rows, n), device=x.device, dtype=x.dtype)
    geo = dict(B=rows, Hs=1, Ws=1, Cs=k, Hd=1, Wd=1, Cd=n)
    assert res is None or act == ACT_NONE
    ops.conv_gemm(x, w, y, act=act, shift=bias, res=res, **geo)
    if self.tape is not None:

      def bwd(dy):
        self.side.label = f'raw_linear {rows}x{n}x{k}'
        dz = ops.act_bwd(dy, y, act) if act != ACT_NONE else dy
        self.side.run(Tape.current, lambda: (gw(dz, x), gb(dz) if gb is not None else None), dz, x)
        dx = torch.empty((rows, k), device=x.device, dtype=x.dtype)
        ops.conv_gemm(dz, wt, dx, B=rows, Hs=1, Ws=1, Cs=n, Hd=1, Wd=1, Cd=k, mode=1)
        return (dx, dz) if res is not None else dx

      self.rec([y], [x, res] if res is not None else [x], bwd)
    return y

  def layernorm(self, x, ln):
    y, mean, rstd = ops.layernorm_fwd(x, ln.weight.detach(), ln.bias.detach(), ln.eps, save=self.tape is not None)
    if self.tape is not None:

      def bwd(dy):
        # dx alone on the dY chain; dgamma / dbeta only feed the optimizer: weight-gradient lane
        dx = ops.layernorm_bwd(dy, x, ln.weight.detach(), mean, rstd, None, None)
        self.side.label = 'layernorm'
        self.side.in_tail = False
        self.side.run(Tape.current, lambda: ops.layernorm_param_grad(dy, x, mean, rstd, self.g(ln.weight), self.g(ln.bias)), dy, x)
        return dx

      self.rec([y], [x], bwd)
    return y

  def add_layernorm(self, a, b, p_drop, ln):
    """LayerNorm(a + dropout(b)) where the sum has no other consumer (post-norm decoder layers): one launch forward, one on the dY chain in
    backward (+ the parameter gradients on the weight-gradient lane) instead of two and three."""
    p = p_drop if self.training else 0.0
    seed = self.next_seed() if p > 0 else 0
    y, s, mean, rstd = ops.add_layernorm_fwd(a, b, ln.weight.detach(), ln.bias.detach(), ln.eps, p, seed, save=self.tape is not None)
    if self.tape is not None:

      def bwd(dy):
        ds, db = ops.add_layernorm_bwd(dy, s, ln.weight.detach(), mean, rstd, None, None, p, seed)
        self.side.label = 'layernorm'
        self.side.in_tail = False
        self.side.run(Tape.current, lambda: ops.layernorm_param_grad(dy, s, mean, rstd, self.g(ln.weight), self.g(ln.bias)), dy, s)
        return ds, db

      self.rec([y], [a, b], bwd)
    return y

  def add(self, a, b, p_drop=0.0):
    """a + dropout(b)."""
    p = p_drop if self.training else 0.0
    seed = self.next_seed() if p > 0 else 0
    y = ops.add_dropout(a, b, p, seed)
    if self.tape is not None:
      self.rec([y], [a, b], lambda dy: (dy, ops.add_dropout(None, dy, p, seed) if p > 0 else dy))
    return y

  def dropout(self, b, p_drop):
    p = p_drop if self.training else 0.0
    if p <= 0:
      return b
    seed = self.next_seed()
    y = ops.add_dropout(None, b, p, seed)
    if self.tape is not None:
      self.rec([y], [b], lambda dy: ops.add_dropout(None, dy, p, seed))
    return y

  def activation(self, x, act):
    y = ops.affine_act(x, act=act)
    if self.tape is not None:
      self.rec([y], [x], lambda dy: ops.act_bwd(dy, y, act))
    return y

  def add_table(self, x, table, param=None):
    """x + table broadcast over the batch (positional embeddings); ``param``: the nn.Parameter behind table."""
    y = ops.add_bcast(x, table)
    if self.tape is not None:

      def bwd(dy):
        if param is not None and param.requires_grad:
          n = table.numel()
          ops.colsum(dy, self.g(param).view(-1), dy.numel() // n, n, n)
        return dy

      self.rec([y], [x], bwd)
    return y

  def attention(self, q, k, v, B, nh, tq, tk, d, ld_q, ld_kv, scale, p_drop, out_ld):
    """Batched multi-head attention core on head-major slices.  q/k/v are views (with data_ptr offsets) into token
    matrices with row strides ld_q / ld_kv; output [B, tq, nh*d(out_ld)].  The fusion transformers (bf16, 320 tokens) run the
    fused kernels of csrc/attention_kernels.hip: scores, softmax and dropout stay in registers, only the per-row log-sum-exp is
    kept for backward.  The fp32 planning decoder (11 x 65 tokens) keeps batched GEMM + softmax."""
    dev, dt_ = q.device, q.dtype
    p = p_drop if self.training else 0.0
    seed = self.next_seed() if p > 0 else 0
    geo = dict(B=B, nh=nh, T=tq, d=d, ld_q=ld_q, ld_kv=ld_kv, ld_o=out_ld, scale=scale)
    if tq == tk and dt_ == torch.bfloat16 and ops.attn_supported(q, **geo):
      O = torch.empty((B, tq, out_ld), device=dev, dtype=dt_)
      lse = torch.empty(B * nh * tq, device=dev, dtype=F32) if self.tape is not None else None
      ops.attn_fwd(q, k, v, O, lse, p_drop=p, seed=seed, **geo)
      return O, ('fused', lse, O), None, (p, seed)
    if ops.small_attn_supported(tq, tk, d, dt_):  # the planning decoder: one launch per attention (csrc/head_kernels.hip)
      O = torch.empty((B, tq, out_ld), device=dev, dtype=dt_)
      P = torch.empty((B, nh, tq, tk), device=dev, dtype=dt_)
      ops.small_attn_fwd(q, k, v, O, P, B=B, nh=nh, tq=tq, tk=tk, d=d, ld_q=ld_q, ld_kv=ld_kv, ld_o=out_ld, scale=scale, p_drop=p, seed=seed)
      return O, ('small', P), None, (p, seed)
    S = torch.empty((B, nh, tq, tk), device=dev, dtype=dt_)
    ops.bgemm(q, k, S, M=tq, N=tk, K=d, lda=ld_q, ldb=ld_kv, ldc=tk, batch0=B, batch1=nh, a_bs=(tq * ld_q, d), b_bs=(tk * ld_kv, d),
              c_bs=(nh * tq * tk, tq * tk))
    P, Pd = ops.softmax_fwd(S, B * nh * tq, tk, tk, alpha=scale, p_drop=p, seed=seed)
    O = torch.empty((B, tq, out_ld), device=dev, dtype=dt_)
    ops.bgemm(Pd, v, O, M=tq, N=d, K=tk, lda=tk, ldb=ld_kv, ldc=out_ld, batch0=B, batch1=nh, a_bs=(nh * tq * tk, tq * tk),
              b_bs=(tk * ld_kv, d), c_bs=(tq * out_ld, d), b_km=True)
    return O, P, Pd, (p, seed)

  def attention_bwd(self, dO, q, k, v, dq, dk, dv, P, Pd, drop, B, nh, tq, tk, d, ld_q, ld_kv, scale, out_ld):
    """Writes dq/dk/dv (views with the same strides as q/k/v)."""
    p, seed = drop
    dev, dt_ = dO.device, dO.dtype
    if isinstance(P, tuple) and P[0] == 'fused':
      _, lse, O = P
      delta = torch.empty(B * nh * tq, device=dev, dtype=F32)
      ops.attn_bwd(q, k, v, O, lse, dO, dq, dk, dv, delta, B=B, nh=nh, T=tq, d=d, ld_q=ld_q, ld_kv=ld_kv, ld_o=out_ld, scale=scale, p_drop=p,
                   seed=seed)
      return
    if isinstance(P, tuple) and P[0] == 'small':
      ops.small_attn_bwd(q, k, v, P[1], dO, dq, dk, dv, B=B, nh=nh, tq=tq, tk=tk, d=d, ld_q=ld_q, ld_kv=ld_kv, ld_o=out_ld, scale=scale, p_drop=p,
                         seed=seed)
      return
    # dV = Pd^T dO
    ops.bgemm(Pd, dO, dv, M=tk, N=d, K=tq, lda=tk, ldb=out_ld, ldc=ld_kv, batch0=B, batch1=nh, a_bs=(nh * tq * tk, tq * tk),
              b_bs=(tq * out_ld, d), c_bs=(tk * ld_kv, d), a_km=True, b_km=True)
    dP = torch.empty((B, nh, tq, tk), device=dev, dtype=dt_)
    ops.bgemm(dO, v, dP, M=tq, N=tk, K=d, lda=out_ld, ldb=ld_kv, ldc=tk, batch0=B, batch1=nh, a_bs=(tq * out_ld, d),
              b_bs=(tk * ld_kv, d), c_bs=(nh * tq * tk, tq * tk))
    ops.softmax_bwd(P, dP, B * nh * tq, tk, tk, alpha=scale, p_drop=p, seed=seed)
    ops.bgemm(dP, k, dq, M=tq, N=d, K=tk, lda=tk, ldb=ld_kv, ldc=ld_q, batch0=B, batch1=nh, a_bs=(nh * tq * tk, tq * tk),
              b_bs=(tk * ld_kv, d), c_bs=(tq * ld_q, d), b_km=True)
    ops.bgemm(dP, q, dk, M=tk, N=d, K=tq, lda=tk, ldb=ld_q, ldc=ld_kv, batch0=B, batch1=nh, a_bs=(nh * tq * tk, tq * tk),
              b_bs=(tq * ld_q, d), c_bs=(tk * ld_kv, d), a_km=True, b_km=True)

  # ------------------------------------------------------------------------------------------------ RegNet
  def bottleneck(self, x, key, blk):
    """RegNet-Y block (timm Bottleneck).  Round 6: conv1 and conv2 hand their consumers the raw convolution output plus
    the BatchNorm statistics (BnView): conv2 normalises conv1's output while staging its halo tiles, the squeeze-excite passes normalise
    conv2's; the only normalised tensors written are the gated input of conv3 and the block output."""
    sc = self.conv(x, key + '.downsample') if blk.downsample is not None else x
    y = self.conv(x, key + '.conv1', act=ACT_RELU, lazy=True)
    y = self.conv(y, key + '.conv2', act=ACT_RELU, lazy=True)
    y = self.squeeze_excite(y, blk.se)
    return self.conv(y, key + '.conv3', act=ACT_RELU, res=sc)

  def squeeze_excite(self, x, se):
    B, H, W, C = x.shape
    w1, b1 = se.fc1.weight.detach().view(se.fc1.weight.shape[0], C), se.fc1.bias.detach()
    w2, b2 = se.fc2.weight.detach().view(C, -1), se.fc2.bias.detach()
    if isinstance(x, BnView) and (not x.relu or B > 64):
      x = self.materialize(x)
    if isinstance(x, BnView):
      # x = relu(BN2(raw2)) exists only as (raw2, statistics): squeeze from raw2 (finalize prologue), then ONE pass writes the gated tensor
      raw, sL = x.raw, x.spec
      pool = ops.mean_hw_bn(raw, self._bn_desc(x))
      hidden, gate = ops.se_gate_fwd(pool, w1, b1, w2, b2)
      y = ops.bn_apply_rows(raw, self._bn_desc(x), gate=gate, rows_per_batch=H * W, relu_pre=True)
      if self.tape is not None:

        def bwd_view(dy):
          # dgate * gate = sum_hw dy * y with the gated tensor y as conv3 read it (the gate's gradient is a residual of cancelling sums: it
          # has to be taken on the values the forward pass used, roundings included -- tfpp.h tfpp_se_gate_bwd_premul)
          dgate_g = ops.se_dgate(dy, y)
          dpool = ops.se_gate_bwd(dgate_g, gate, hidden, pool, w1, w2, self.g(se.fc1.weight), self.g(se.fc1.bias),
                                  self.g(se.fc2.weight), self.g(se.fc2.bias), premul=True)
          # dx is the complete gradient of relu(BN2(raw2)) (its only consumer is this block): the BatchNorm-backward sums come with it
          dx, partial, nrows = ops.se_bwd_apply_bn(dy, gate, dpool, raw, sL.scale, sL.shift, sL.save_mean, sL.save_invstd)
          if partial is not None:
            self._bn_pre[_key(x)] = (partial, nrows, dx)
          return dx

        self.rec([y], [x], bwd_view)
      return y
    pool = ops.mean_hw(x)
    hidden, gate = ops.se_gate_fwd(pool, w1, b1, w2, b2)
    y = ops.affine_act(x, gate=gate, rows_per_batch=H * W)
    if self.tape is not None:

      def bwd(dy):
        dgate = ops.se_dgate(dy, x)
        dpool = ops.se_gate_bwd(dgate, gate, hidden, pool, w1, w2, self.g(se.fc1.weight), self.g(se.fc1.bias),
                                self.g(se.fc2.weight), self.g(se.fc2.bias))
        info = self._bn_of.get(_key(x)) if x.dtype == torch.bfloat16 else None
        if info is not None and info[2] and Tape.current.is_last_contribution(x) and not (self.sync_bn and self.sync_world > 1):
          # x = relu(BN(raw)) of conv2: this IS its complete gradient -> emit the BatchNorm-backward sums in the same pass
          sL, rawL, _ = info
          dx, partial, nrows = ops.se_bwd_apply_bns(dy, gate, dpool, x, rawL, sL.save_mean, sL.save_invstd)
          self._bn_pre[_key(x)] = (partial, nrows, dx)
          return dx
        return ops.se_bwd_apply(dy, gate, dpool)

      self.rec([y], [x], bwd)
    return y

  def stage(self, x, key, stage_mod):
    for bname, blk in stage_mod.named_children():
      x = self.bottleneck(x, f'{key}.{bname}', blk)
    return x

  # ------------------------------------------------------------------------------------------------ fusion GPT
  def pool_tokens(self, x, ho, wo):
    B, H, W, C = x.shape
    y = ops.avgpool_fwd(x, ho, wo)
    if self.tape is not None:

      def bwd(dy):
        dx = ops.zeros((B, H, W, C), x.dtype, x.device)
        ops.avgpool_bwd_add(dy.view(y.shape), dx, ho, wo)  # (a consumer may have recorded a reshaped view: gradients are keyed by storage)
        return dx

      self.rec([y], [x], bwd)
    return y

  def upsample_add(self, tok, base, mul=None):
    """base + bilinear(tok) (F.interpolate align_corners=False, transfuser.py:239-255)."""
    B, hi, wi, C = tok.shape
    _, ho, wo, _ = base.shape
    y = ops.bilinear_fwd(tok, ho, wo, base=base)
    if self.tape is not None:
      self.rec([y], [tok, base], lambda dy: (ops.bilinear_bwd(dy.view(y.shape), hi, wi), dy))
    return y

  def upsample(self, x, ho, wo, mul=None):
    B, hi, wi, C = x.shape
    y = ops.bilinear_fwd(x, ho, wo, mul=mul)
    if self.tape is not None:
      self.rec([y], [x], lambda dy: ops.bilinear_bwd(dy, hi, wi, mul=mul))
    return y

  def gpt(self, i, img_tok, lid_tok):
    """team_code/transfuser.py:301-339: tokens [B,256,C] + [B,64,C] -> same shapes."""
    g = self.m.backbone.transformers[i]
    cfg = self.cfg
    B, ni, C = img_tok.shape[0], img_tok.shape[1] * img_tok.shape[2], img_tok.shape[3]
    nl = lid_tok.shape[1] * lid_tok.shape[2]
    T = ni + nl
    dt_ = img_tok.dtype
    x0 = torch.empty((B, T, C), device=img_tok.device, dtype=dt_)
    ops.copy_rows(img_tok, x0, B, ni * C, ni * C, 0, T * C, 0)
    ops.copy_rows(lid_tok, x0, B, nl * C, nl * C, 0, T * C, ni * C)
    if self.tape is not None:

      def bwd_cat(d):
        di = torch.empty_like(img_tok)
        dl = torch.empty_like(lid_tok)
        ops.copy_rows(d, di, B, ni * C, T * C, 0, ni * C, 0)
        ops.copy_rows(d, dl, B, nl * C, T * C, ni * C, nl * C, 0)
        return di, dl

      self.rec([x0], [img_tok, lid_tok], bwd_cat)
    x = self.add_table(x0, g.pos_emb.detach().view(-1), g.pos_emb)
    x = self.dropout(x, cfg.embd_pdrop)
    for l, blk in enumerate(g.blocks):
      x = self.gpt_block(i, l, blk, x, B, T, C)
    x = self.layernorm(x, g.ln_f)
    io = torch.empty((B, img_tok.shape[1], img_tok.shape[2], C), device=x.device, dtype=dt_)
    lo = torch.empty((B, lid_tok.shape[1], lid_tok.shape[2], C), device=x.device, dtype=dt_)
    ops.copy_rows(x, io, B, ni * C, T * C, 0, ni * C, 0)
    ops.copy_rows(x, lo, B, nl * C, T * C, ni * C, nl * C, 0)
    if self.tape is not None:

      def bwd_split(di, dl):
        d = ops.zeros((B, T, C), dt_, x.device)
        if di is not None:
          ops.copy_rows(di, d, B, ni * C, ni * C, 0, T * C, 0)
        if dl is not None:
          ops.copy_rows(dl, d, B, nl * C, nl * C, 0, T * C, ni * C)
        return d

      self.rec([io, lo], [x], bwd_split)
    return io, lo

  def gpt_block(self, i, l, blk, x, B, T, C):
    cfg = self.cfg
    st = self._attn_state(i, l)
    nh, d, dp = st['nh'], st['d'], st['dp']
    npk = 3 * nh * dp
    key = f'backbone.transformers.{i}.blocks.{l}'
    h = self.layernorm(x, blk.ln1)

    def gw_qkv(dz, xin):
      for j, lin in enumerate((blk.attn.query, blk.attn.key, blk.attn.value)):
        dslice = dz.view(-1, npk)[:, j * nh * dp:]
        ops.conv_wgrad(dslice, xin, self.g(lin.weight), B=B * T, Hs=1, Ws=1, Cs=C, Hd=1, Wd=1, Cd=nh * dp, c_real=C,
                       row_map=st['rmap'], dy_ld=npk, dw_ld=C)

    def gb_qkv(dz):
      tmp = ops.zeros(npk, F32, dz.device)
      ops.colsum(dz, tmp, B * T, npk, npk)
      un = torch.empty(3 * C, device=dz.device, dtype=F32)
      for j, lin in enumerate((blk.attn.query, blk.attn.key, blk.attn.value)):
        ops.pack2d(tmp[j * nh * dp:], un[j * C:], 1, C, nh * dp, C, col_map=st['inv'])
        ops.copy_rows(un[j * C:], self.g(lin.bias), 1, C, 0, 0, 0, 0, accumulate=True)  # every gradient kernel ADDS to the arena (gradient accumulation)

    qkv = self.raw_linear(h.view(B * T, C), st['wqkv'], st['bqkv'], st.get('wqkv_t'), gw_qkv, gb_qkv, npk, C)
    q, k, v = qkv.view(-1)[0:], qkv.view(-1)[nh * dp:], qkv.view(-1)[2 * nh * dp:]
    O, P, Pd, drop = self.attention(q, k, v, B, nh, T, T, dp, npk, npk, 1.0 / math.sqrt(d), cfg.attn_pdrop, nh * dp)
    if self.tape is not None:

      def bwd_attn(dO):
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv.view(-1)[0:], dqkv.view(-1)[nh * dp:], dqkv.view(-1)[2 * nh * dp:]
        self.attention_bwd(dO, q, k, v, dq, dk, dv, P, Pd, drop, B, nh, T, T, dp, npk, npk, 1.0 / math.sqrt(d), nh * dp)
        return dqkv

      self.rec([O], [qkv], bwd_attn)

    def gw_proj(dz, xin):
      ops.conv_wgrad(dz, xin, self.g(blk.attn.proj.weight), B=B * T, Hs=1, Ws=1, Cs=nh * dp, Hd=1, Wd=1, Cd=C, c_real=nh * dp,
                     col_map=st['rmap'], dw_ld=C)

    def gb_proj(dz):
      ops.colsum(dz, self.g(blk.attn.proj.bias), B * T, C, C)

    # residual adds: with dropout between the linear and the sum (training) a separate add_dropout launch; without one (the 20 Hz inference
    # tick, resid_pdrop = 0) the sum is the residual operand of the GEMM epilogue -- two launches fewer per block on a launch-bound chain
    fuse_res = not (self.training and cfg.resid_pdrop > 0)
    y = self.raw_linear(O.view(B * T, nh * dp), st['wproj'], blk.attn.proj.bias.detach(), st.get('wproj_t'), gw_proj, gb_proj, C,
                        nh * dp, res=x.view(B * T, C) if fuse_res else None)
    x = y.view(B, T, C) if fuse_res else self.add(x, y.view(B, T, C), cfg.resid_pdrop)
    h = self.layernorm(x, blk.ln2)
    h = self.linear(h, key + '.mlp.0', act=ACT_RELU)
    if fuse_res:
      return self.linear(h, key + '.mlp.2', res=x)
    h = self.linear(h, key + '.mlp.2')
    return self.add(x, h, cfg.resid_pdrop)

  # ------------------------------------------------------------------------------------------------ planning head (fp32)
  def mha(self, x, mem, prefix, layer_attn, B, tq, tk, self_attn):
    """nn.MultiheadAttention (batch_first) as nn.TransformerDecoderLayer uses it (model.py:137-143)."""
    cfg = self.cfg
    dm, nh = x.shape[-1], cfg.num_decoder_heads
    d = dm // nh
    pd = self._decoder_dropout()
    w, b = layer_attn.in_proj_weight, layer_attn.in_proj_bias
    wd, bd = w.detach(), b.detach()

    def proj(inp, r0, r1, rows):
      n = r1 - r0

      def gw(dz, xin):
        ops.conv_wgrad(dz, xin, self.g(w)[r0:r1], B=rows, Hs=1, Ws=1, Cs=dm, Hd=1, Wd=1, Cd=n, c_real=dm, dw_ld=dm)

      def gb(dz):
        ops.colsum(dz, self.g(b)[r0:r1], rows, n, n)

      y = torch.empty((rows, n), device=inp.device, dtype=F32)
      ops.conv_gemm(inp, wd[r0:r1], y, B=rows, Hs=1, Ws=1, Cs=dm, Hd=1, Wd=1, Cd=n, shift=bd[r0:r1])
      if self.tape is not None:

        def bwd(dy):
          self.side.run(Tape.current, lambda: (gw(dy, inp), gb(dy)), dy, inp)
          dx = torch.empty((rows, dm), device=inp.device, dtype=F32)
          # dx = dy @ W[r0:r1]  : W slice [n, dm] is k-major for this product -> batched GEMM with b_km
          ops.bgemm(dy, wd[r0:r1], dx, M=rows, N=dm, K=n, lda=n, ldb=dm, ldc=dm, b_km=True)
          return dx

        self.rec([y], [inp], bwd)
      return y

    if self_attn:
      qkv = proj(x.view(B * tq, dm), 0, 3 * dm, B * tq)
      q, k, v = qkv.view(-1)[0:], qkv.view(-1)[dm:], qkv.view(-1)[2 * dm:]
      ld_q = ld_kv = 3 * dm
      srcs = [qkv]
    else:
      qq = proj(x.view(B * tq, dm), 0, dm, B * tq)
      kv = proj(mem.view(B * tk, dm), dm, 3 * dm, B * tk)
      q, k, v = qq.view(-1), kv.view(-1)[0:], kv.view(-1)[dm:]
      ld_q, ld_kv = dm, 2 * dm
      srcs = [qq, kv]
    O, P, Pd, drop = self.attention(q, k, v, B, nh, tq, tk, d, ld_q, ld_kv, 1.0 / math.sqrt(d), pd, dm)
    if self.tape is not None:

      def bwd_attn(dO):
        if self_attn:
          dqkv = torch.empty_like(srcs[0])
          dq, dk, dv = dqkv.view(-1)[0:], dqkv.view(-1)[dm:], dqkv.view(-1)[2 * dm:]
          self.attention_bwd(dO, q, k, v, dq, dk, dv, P, Pd, drop, B, nh, tq, tk, d, ld_q, ld_kv, 1.0 / math.sqrt(d), dm)
          return dqkv
        dqq, dkv = torch.empty_like(srcs[0]), torch.empty_like(srcs[1])
        self.attention_bwd(dO, q, k, v, dqq.view(-1), dkv.view(-1)[0:], dkv.view(-1)[dm:], P, Pd, drop, B, nh, tq, tk, d, ld_q, ld_kv,
                           1.0 / math.sqrt(d), dm)
        return dqq, dkv

      self.rec([O], srcs, bwd_attn)
    return self.linear(O, prefix + '.out_proj')

  def _decoder_dropout(self):
    return float(self.m.join.layers[0].dropout.p)

  def decoder(self, query, mem, B, tq, tk):
    """nn.TransformerDecoder: post-norm layers, ReLU FFN as actually run (see DESIGN.md 'decoder activation')."""
    pd = self._decoder_dropout()
    x = query
    for l, layer in enumerate(self.m.join.layers):
      q = f'join.layers.{l}'
      x = self.add_layernorm(x, self.mha(x, None, q + '.self_attn', layer.self_attn, B, tq, tq, True), pd, layer.norm1)
      x = self.add_layernorm(x, self.mha(x, mem, q + '.multihead_attn', layer.multihead_attn, B, tq, tk, False), pd, layer.norm2)
      h = self.linear(x, q + '.linear1', act=ACT_RELU)
      h = self.dropout(h, pd)
      h = self.linear(h, q + '.linear2')
      x = self.add_layernorm(x, h, pd, layer.norm3)
    return self.layernorm(x, self.m.join.norm)

  def attention_with_weights(self, x, src, prefix, mod, B, tq, tk, acc=None):
    """team_code/transfuser.py:404-443 (MultiheadAttentionWithAttention): query / key / value / proj linears, dropout on the probabilities and on the
    projected output.  ``acc`` [nh, tq, tk]: the probabilities of sample 0 (before dropout) are added to it -- the attention read-out of tp_attention."""
    dm, nh = x.shape[-1], self.cfg.num_decoder_heads
    d = dm // nh
    q = self.linear(x, prefix + '.query')
    k = self.linear(src, prefix + '.key')
    v = self.linear(src, prefix + '.value')
    scale = 1.0 / math.sqrt(d)
    O, P, Pd, drop = self.attention(q.view(-1), k.view(-1), v.view(-1), B, nh, tq, tk, d, dm, dm, scale, float(mod.attn_drop.p), dm)
    if acc is not None:
      probs = P[1] if isinstance(P, tuple) else P  # [B, nh, tq, tk]
      ops.copy_rows(probs, acc, 1, nh * tq * tk, 0, 0, 0, 0, accumulate=True)
    if self.tape is not None:

      def bwd_attn(dO):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        self.attention_bwd(dO, q.view(-1), k.view(-1), v.view(-1), dq.view(-1), dk.view(-1), dv.view(-1), P, Pd, drop, B, nh, tq, tk, d, dm, dm, scale, dm)
        return dq, dk, dv

      self.rec([O], [q, k, v], bwd_attn)
    return self.dropout(self.linear(O, prefix + '.proj'), float(mod.resid_drop.p))

  def decoder_with_attention(self, query, mem, B, tq, tk, acc):
    """team_code/transfuser.py:447-508 (TransformerDecoderWithAttention, the decoder of config.tp_attention): post-norm layers with the exact GELU in
    the FFN (here the activation module survives the per-layer deep copy) and a final LayerNorm."""
    pd = self._decoder_dropout()
    x = query
    for l, layer in enumerate(self.m.join.layers):
      q = f'join.layers.{l}'
      x = self.add_layernorm(x, self.attention_with_weights(x, x, q + '.self_attn', layer.self_attn, B, tq, tq), pd, layer.norm1)
      x = self.add_layernorm(x, self.attention_with_weights(x, mem, q + '.multihead_attn', layer.multihead_attn, B, tq, tk, acc), pd, layer.norm2)
      h1 = self.linear(x, q + '.linear1')
      h = ops.affine_act(h1, act=ACT_GELU)
      if self.tape is not None:
        self.rec([h], [h1], lambda dh, h1=h1: ops.act_bwd(dh, h1, ACT_GELU))  # (the exact GELU derivative needs the pre-activation)
      h = self.dropout(h, pd)
      h = self.linear(h, q + '.linear2')
      x = self.add_layernorm(x, h, pd, layer.norm3)
    return self.layernorm(x, self.m.join.norm)

  def gru_decoder(self, feats, target_point, dec, name, B, T):
    """team_code/model.py:857-867."""
    dm = feats.shape[-1]
    tp4 = ops.zeros((B, 4), F32, feats.device)
    ops.copy_rows(target_point, tp4, B, 2, 2, 0, 4, 0)
    h0 = self.linear(tp4, name + '.encoder', x_grad=False)
    gi = self.linear(feats, name + '.gru.ih')
    whh, bhh = dec.gru.weight_hh_l0, dec.gru.bias_hh_l0
    wdec, bdec = dec.decoder.weight, dec.decoder.bias
    out, save = ops.gru_fwd(gi, h0, whh.detach(), bhh.detach(), wdec.detach(), bdec.detach())
    if self.tape is not None:

      def bwd(dout):
        # the recurrence's BPTT is the first node of the planning head's backward chain: only dgi / dh0 are produced on it, the per-sample
        # partial images of the parameter gradients are summed on the weight-gradient lane
        dgi, dh0, part, reduce = ops.gru_bwd(dout, save, h0, whh.detach(), bhh.detach(), wdec.detach(), self.g(whh), self.g(bhh), self.g(wdec),
                                             self.g(bdec), defer=True)
        self.side.label = name + '.gru'
        self.side.in_tail = False
        self.side.run(Tape.current, reduce, part)
        return dgi, dh0

      self.rec([out], [gi, h0], bwd)
    return out

  # ------------------------------------------------------------------------------------------------ full forward
  def forward(self, rgb, lidar_bev, target_point, ego_vel, command):
    """Returns a dict of internal tensors (NHWC, channel-padded) -- see model.py for the caller-facing tuple."""
    m, cfg, dt_ = self.m, self.cfg, self.dtype
    dev = rgb.device
    B = rgb.shape[0]
    bb = m.backbone
    out = {}
    self._bn_of, self._bn_pre = {}, {}
    self._relu_of, self._relu_done = {}, {}
    if self.sync_bn:
      import torch.distributed as dist
      self.sync_world = dist.get_world_size(self.sync_group) if (dist.is_available() and dist.is_initialized()) else 1
      if self.training and self.sync_world > 1 and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('carla_garage_amd: a SyncBatchNorm training step puts collectives inside the pass and cannot be captured into a hipGraph '
                           '(use Trainer.train_step / the eager drop-in step: TFPP_DROPIN_GRAPH_AFTER=-1)')
    if ops.NODE_HASH['on'] and self.tape is not None:
      ops.node_hash_begin(dev)
    if self.tape is not None:
      self.tape.on_accumulate = lambda key: (self._bn_pre.pop(key, None), self._relu_done.pop(key, None))  # a "complete" gradient got another addend: sums / mask are stale
    self.lanes.begin(dev)
    mul = add = None
    if cfg.normalize_imagenet:
      mul = self._const('img_mul', lambda: torch.tensor([1.0 / (255.0 * s) for s in (0.229, 0.224, 0.225)]))
      add = self._const('img_add', lambda: torch.tensor([-mu / s for mu, s in ((0.485, 0.229), (0.456, 0.224), (0.406, 0.225))]))
    if rgb.dtype == torch.uint8:
      # the frame as the caller holds it: [B,H,W,3] = cv2's HWC BGR (sensor_agent.py:277-286 after imdecode), [B,3,H,W] = the loader's RGB CHW
      hwc = rgb.shape[-1] == 3 and rgb.shape[1] != 3
      xi = ops.u8_to_nhwc_affine(rgb.contiguous(), dt_, 8, mul, add, hwc=hwc, swap=hwc)
    else:
      xi = ops.nchw_to_nhwc_affine(rgb.float().contiguous(), dt_, 8, mul, add)
    lanes = self.lanes
    # two-stage training (train.py:495-508 freeze_backbone): a frozen backbone is not walked by backward, and neither is a frozen head that
    # only consumes its features
    dead_feats = self.tape is not None and self._frozen(bb)
    n_backbone = len(self.tape.nodes) if self.tape is not None else 0
    if self.aim or self.bev or self.video:
      self._join_pack()  # (only the default TransFuser backbone is laid out for the deferred repack)
    if self.aim:  # team_code/aim.py:32-61: the image branch alone; fused_features = the stage-4 feature grid
      xi = self.conv(xi, 'backbone.image_encoder.stem', act=ACT_RELU, x_grad=False)
      for i in range(4):
        xi = self.stage(xi, f'backbone.image_encoder.s{i + 1}', bb.image_encoder[f's{i + 1}'])
      xl = xi
    elif self.bev:  # team_code/bev_encoder.py:146-233
      xi, xl = self.bev_runner.forward(xi, lidar_bev.float().contiguous())
    else:
      lidar_in = lidar_bev.float().contiguous()
      lanes.hold(lidar_in)
      nt = bb.lidar_time_frames if self.video else 1  # time frames of the LiDAR feature maps (transfuser.py:50)
      with lanes.fork():  # the LiDAR branch runs on its own stream between the fusion points
        if self.video:
          xl = self.swin.stem(lidar_in)  # [B, 3, 64, 64, 96]
        else:
          xl = ops.nchw_to_nhwc_affine(lidar_in, dt_, 8)
          xl = self.conv(xl, 'backbone.lidar_encoder.stem', act=ACT_RELU, x_grad=False)
      xi = self.conv(xi, 'backbone.image_encoder.stem', act=ACT_RELU, x_grad=False)
      for i in range(4):
        n_lidar = len(self.tape.nodes) if self.tape is not None else 0
        with lanes.fork():
          if self.video:
            xl = self.swin.layer(i, xl)  # [B, 3, H, W, C]
            xl = xl.view(B * nt, xl.shape[2], xl.shape[3], xl.shape[4])  # the time frames as batch entries for the 2-D pooling / resampling
          else:
            xl = self.stage(xl, f'backbone.lidar_encoder.s{i + 1}', bb.lidar_encoder[f's{i + 1}'])
          lt = self.pool_tokens(xl, cfg.lidar_vert_anchors, cfg.lidar_horz_anchors)
          lt = self.conv(lt, f'backbone.lidar_channel_to_img.{i}')
          lt = lt.view(B, nt * lt.shape[1], lt.shape[2], lt.shape[3])  # tokens in (t, h, w) order (transfuser.py:319)
        if i == 0 and self.tape is not None and not self.video:
          # backward of LiDAR stage 1 on lane 0, behind image stage 1: in the captured step lane 1 does not get to run it before the
          # weight-gradient batch in flight has drained (tools/lane_timeline.py: 2.4 ms after its inputs are ready), which delays the last
          # weight-gradient batch -- the tail of the step -- by as much
          self.tape.relane(n_lidar, len(self.tape.nodes), 0)
        xi = self.stage(xi, f'backbone.image_encoder.s{i + 1}', bb.image_encoder[f's{i + 1}'])
        it = self.pool_tokens(xi, cfg.img_vert_anchors, cfg.img_horz_anchors)
        lanes.join()
        self._join_pack()  # (first fusion point: from here on the layers use the weight images of the deferred part of the repack)
        io, lo = self.gpt(i, it, lt)
        lanes.hold(lt, lo)
        with lanes.fork():
          lo = self.conv(lo.view(B * nt, lo.shape[1] // nt, lo.shape[2], lo.shape[3]), f'backbone.img_channel_to_lidar.{i}')
          xl = self.upsample_add(lo, xl)  # trilinear with an unchanged time axis = bilinear per frame (transfuser.py:243-248)
          if self.video:
            xl = xl.view(B, nt, xl.shape[1], xl.shape[2], xl.shape[3])
        xi = self.upsample_add(io, xi)
      if self.video:  # transfuser.py:176-180: average the remaining time frames
        with lanes.fork():
          _, _, hh_, ww_, cc_ = xl.shape
          acc = ops.zeros((B, hh_, ww_, cc_), dt_, dev)
          for t_ in range(nt):
            ops.copy_rows(xl, acc, B, hh_ * ww_ * cc_, nt * hh_ * ww_ * cc_, t_ * hh_ * ww_ * cc_, hh_ * ww_ * cc_, 0, accumulate=True)
          third = self._const(f'time_mean{nt}x{cc_}', lambda: torch.full((cc_,), 1.0 / nt))
          zero_c = self._const(f'zeros{cc_}', lambda: torch.zeros(cc_))
          xl5 = xl
          xl = ops.affine_act(acc, scale=third, shift=zero_c)
          if self.tape is not None:

            def bwd_mean(dm, nt=nt, n=hh_ * ww_ * cc_, shape=tuple(xl5.shape)):
              ds = ops.affine_act(dm.view(shape[0], shape[2], shape[3], shape[4]), scale=third, shift=zero_c)
              dx5 = torch.empty(shape, device=dm.device, dtype=dm.dtype)
              for t_ in range(nt):
                ops.copy_rows(ds, dx5, shape[0], n, n, 0, nt * n, t_ * n)
              return dx5

            self.rec([xl], [xl5], bwd_mean)
      lanes.join()
    out['image_feature_grid'], out['fused_features'] = xi, xl
    if dead_feats:
      self._drop_dead_nodes(n_backbone)

    # planning head, fp32 (model.py:299-358)
    # on the LiDAR lane (idle after the backbone): ~250 tiny latency-bound launches that overlap with the dense heads below
    hoist_from = len(self.tape.nodes) if self.tape is not None else 0
    with self.lanes.fork():
      dm = cfg.gru_input_size
      x = self.conv(xl, 'change_channel', out_f32=True, x_grad=not dead_feats)  # [B,8,8,256] fp32
      hh, ww = x.shape[1], x.shape[2]
      pos = self._const(f'sine{hh}x{ww}', lambda: m.sine_table(hh, ww))
      x = self.add_table(x, pos.view(-1))
      vn = m.velocity_normalization
      ev = ego_vel.float().contiguous()
      if self.sync_bn and self.training and vn.training and self.sync_world > 1:
        # the BatchNorm1d on the ego speed is converted too: normalise with the statistics of every rank's speeds (gather, normalise the
        # gathered vector -- identical on every rank, as are the running statistics it updates -- keep the own slice)
        import torch.distributed as dist
        parts = [torch.empty_like(ev) for _ in range(self.sync_world)]
        dist.all_gather(parts, ev, group=self.sync_group)
        allv = torch.empty((self.sync_world * B, 1), device=dev, dtype=F32)
        for k, pt in enumerate(parts):
          ops.copy_rows(pt, allv, 1, B, 0, 0, 0, k * B)
        r = dist.get_rank(self.sync_group)
        vel = ops.bn1d_scalar(allv, vn.running_mean, vn.running_var, vn.num_batches_tracked, True, vn.momentum, vn.eps)[r * B:(r + 1) * B].contiguous()
      else:
        vel = ops.bn1d_scalar(ev, vn.running_mean, vn.running_var, vn.num_batches_tracked, self.training and vn.training, vn.momentum, vn.eps)
      es_in = ops.zeros((B, 8), F32, dev)
      ops.copy_rows(vel, es_in, B, 1, 1, 0, 8, 0)
      ops.copy_rows(command.float().contiguous(), es_in, B, 6, 6, 0, 8, 1)
      es = self.linear(es_in, 'extra_sensor_encoder.0', act=ACT_RELU, x_grad=False)
      es = self.linear(es, 'extra_sensor_encoder.2', act=ACT_RELU)
      es = self.add_table(es, m.extra_sensor_pos_embed.detach().view(-1), m.extra_sensor_pos_embed)
      ntok = hh * ww
      extra = [es]
      if self.tp_attention:  # model.py:336-339: the encoded target point is one more memory token
        tp4 = ops.zeros((B, 4), F32, dev)
        ops.copy_rows(target_point.float().contiguous(), tp4, B, 2, 2, 0, 4, 0)
        tpt = self.linear(self.linear(tp4, 'tp_encoder.0', act=ACT_RELU, x_grad=False), 'tp_encoder.2')
        extra.append(self.add_table(tpt, m.tp_pos_embed.detach().view(-1), m.tp_pos_embed))
      nmem = ntok + len(extra)
      mem = torch.empty((B, nmem, dm), device=dev, dtype=F32)
      ops.copy_rows(x, mem, B, ntok * dm, ntok * dm, 0, nmem * dm, 0)
      for i_, e_ in enumerate(extra):
        ops.copy_rows(e_, mem, B, dm, dm, 0, nmem * dm, (ntok + i_) * dm)
      if self.tape is not None:

        def bwd_mem(d):
          dx_ = torch.empty_like(x)
          ops.copy_rows(d, dx_, B, ntok * dm, nmem * dm, 0, ntok * dm, 0)
          des = [torch.empty_like(e_) for e_ in extra]
          for i_, de in enumerate(des):
            ops.copy_rows(d, de, B, dm, nmem * dm, (ntok + i_) * dm, dm, 0)
          return (dx_,) + tuple(des)

        self.rec([mem], [x] + extra, bwd_mem)
      out['memory'] = mem

      def run_queries(qparam, nq):
        q0 = torch.empty((B, nq, dm), device=dev, dtype=F32)
        ops.copy_rows(qparam.detach(), q0, B, nq * dm, 0, 0, nq * dm, 0)
        if self.tape is not None:

          def bwd_q(d):
            if qparam.requires_grad:
              ops.colsum(d, self.g(qparam).view(-1), B, nq * dm, nq * dm)
            return ()

          self.rec([q0], [], bwd_q)
        if self.tp_attention:
          out['attn_acc'] = ops.zeros((cfg.num_decoder_heads, nq, nmem), F32, dev)
          out['attn_layers'] = len(m.join.layers)
          return self.decoder_with_attention(q0, mem, B, nq, nmem, out['attn_acc'])
        return self.decoder(q0, mem, B, nq, nmem)

      out['pred_wp'] = out['pred_target_speed'] = out['pred_checkpoint'] = None
      if cfg.use_wp_gru and getattr(m, 'multi_wp', False):
        # multi_wp_output (model.py:326-331): 2 nq + 1 queries -> two GRU decoders (one hypothesis each) and the path-selection logit
        nq = cfg.pred_len // cfg.wp_dilation
        nqa = 2 * nq + 1
        j = run_queries(m.wp_query, nqa)
        g0, g1 = (torch.empty((B, nq, dm), device=dev, dtype=F32) for _ in range(2))
        sf = torch.empty((B, dm), device=dev, dtype=F32)
        ops.copy_rows(j, g0, B, nq * dm, nqa * dm, 0, nq * dm, 0)
        ops.copy_rows(j, g1, B, nq * dm, nqa * dm, nq * dm, nq * dm, 0)
        ops.copy_rows(j, sf, B, dm, nqa * dm, 2 * nq * dm, dm, 0)
        if self.tape is not None:

          def bwd_split(d0, d1, ds):
            d = ops.zeros((B, nqa, dm), F32, dev)
            if d0 is not None:
              ops.copy_rows(d0, d, B, nq * dm, nq * dm, 0, nqa * dm, 0)
            if d1 is not None:
              ops.copy_rows(d1, d, B, nq * dm, nq * dm, 0, nqa * dm, nq * dm)
            if ds is not None:
              ops.copy_rows(ds, d, B, dm, dm, 0, nqa * dm, 2 * nq * dm)
            return d

          self.rec([g0, g1, sf], [j], bwd_split)
        tp = target_point.float().contiguous()
        w0 = self.gru_decoder(g0, tp, m.wp_decoder, 'wp_decoder', B, nq)
        w1 = self.gru_decoder(g1, tp, m.wp_decoder_1, 'wp_decoder_1', B, nq)
        pair = torch.empty((B, 2, nq, 2), device=dev, dtype=F32)  # both hypotheses of a sample side by side: one loss kernel, one gradient seed
        ops.copy_rows(w0, pair, B, nq * 2, nq * 2, 0, nq * 4, 0)
        ops.copy_rows(w1, pair, B, nq * 2, nq * 2, 0, nq * 4, nq * 2)
        if self.tape is not None:

          def bwd_pair(d):
            d0, d1 = torch.empty_like(w0), torch.empty_like(w1)
            ops.copy_rows(d, d0, B, nq * 2, nq * 4, 0, nq * 2, 0)
            ops.copy_rows(d, d1, B, nq * 2, nq * 4, nq * 2, nq * 2, 0)
            return d0, d1

          self.rec([pair], [w0, w1], bwd_pair)
        out['pred_wp_pair'] = pair
        out['selected_path'] = self.linear(sf, 'select_wps')  # [B, 8] (1 real)
      elif cfg.use_wp_gru:
        nq = cfg.pred_len // cfg.wp_dilation
        j = run_queries(m.wp_query, nq)
        out['pred_wp'] = self.gru_decoder(j, target_point.float().contiguous(), m.wp_decoder, 'wp_decoder', B, nq)
      if cfg.use_controller_input_prediction:
        n = cfg.predict_checkpoint_len
        j = run_queries(m.checkpoint_query, n + 1)
        out['joined'] = j
        gf = torch.empty((B, n, dm), device=dev, dtype=F32)
        tsf = torch.empty((B, dm), device=dev, dtype=F32)
        ops.copy_rows(j, gf, B, n * dm, (n + 1) * dm, 0, n * dm, 0)
        ops.copy_rows(j, tsf, B, dm, (n + 1) * dm, n * dm, dm, 0)
        if self.tape is not None:

          def bwd_j(dg, dt2):
            d = ops.zeros((B, n + 1, dm), F32, dev)
            if dg is not None:
              ops.copy_rows(dg, d, B, n * dm, n * dm, 0, (n + 1) * dm, 0)
            if dt2 is not None:
              ops.copy_rows(dt2, d, B, dm, dm, 0, (n + 1) * dm, n * dm)
            return d

          self.rec([gf, tsf], [j], bwd_j)
        out['pred_checkpoint'] = self.gru_decoder(gf, target_point.float().contiguous(), m.checkpoint_decoder, 'checkpoint_decoder',
                                                  B, n)
        ts = self.linear(tsf, 'target_speed_network.0', act=ACT_RELU)
        out['pred_target_speed'] = self.linear(ts, 'target_speed_network.2')  # [B, 8] (4 real)

    if self.tape is not None:
      self.tape.hoist(hoist_from, len(self.tape.nodes))  # backward walks the planning head first (Tape.hoist)
    # BEV feature pyramid (transfuser.py:131-137) with the CenterNet and BEV-semantic heads: 64 x 64 maps, ~25 small launches on lane 0,
    # beside the planning head (lane 1)
    bev = None
    out['pred_bev_semantic'] = None
    out['bb'] = None
    lanes.hold(xl)
    n_bev = len(self.tape.nodes) if self.tape is not None else 0
    if cfg.detect_boxes or cfg.use_bev_semantic:
      p5 = self.conv(xl, 'backbone.c5_conv', act=ACT_RELU)
      p4 = self.upsample(p5, p5.shape[1] * cfg.bev_upsample_factor, p5.shape[2] * cfg.bev_upsample_factor)
      p4 = self.conv(p4, 'backbone.up_conv5', act=ACT_RELU)
      p3 = self.upsample(p4, cfg.lidar_resolution_height // cfg.bev_down_sample_factor,
                         cfg.lidar_resolution_width // cfg.bev_down_sample_factor)
      bev = self.conv(p3, 'backbone.up_conv4', act=ACT_RELU)
    if cfg.use_bev_semantic:
      y = self.conv(bev, 'bev_semantic_decoder.0', act=ACT_RELU)
      y = self.conv(y, 'bev_semantic_decoder.2')
      mask = m.valid_bev_pixels.detach().view(-1)
      out['pred_bev_semantic'] = self.upsample(y, cfg.lidar_resolution_height, cfg.lidar_resolution_width, mul=mask)
    if cfg.detect_boxes:
      bbs = []
      for br in m.head.BRANCHES:
        h = self.conv(bev, f'head.{br}_head.0', act=ACT_RELU)
        bbs.append(self.conv(h, f'head.{br}_head.2', act=ACT_SIGMOID if br == 'heatmap' else ACT_NONE))
      out['bb'] = bbs
    out['bev'] = bev
    if dead_feats and self._frozen(m.head if cfg.detect_boxes else None, m.bev_semantic_decoder if cfg.use_bev_semantic else None):
      self._drop_dead_nodes(n_bev)

    # auxiliary dense heads
    for key, name, on in (('pred_semantic', 'semantic_decoder', cfg.use_semantic), ('pred_depth', 'depth_decoder', cfg.use_depth)):
      n_dec = len(self.tape.nodes) if self.tape is not None else 0
      y = self.perspective_decoder(xi, name) if on else None
      out[key] = self.activation(y, ACT_SIGMOID) if on and key == 'pred_depth' else y
      if on and dead_feats and self._frozen(getattr(m, name)):
        self._drop_dead_nodes(n_dec)
    self.lanes.join()
    return out

  def perspective_decoder(self, x, name):
    """team_code/transfuser_utils.py:697-704."""
    m = getattr(self.m, name)
    x = self.conv(x, name + '.deconv1.0', act=ACT_RELU)
    x = self.conv(x, name + '.deconv1.2', act=ACT_RELU)
    x = self.upsample(x, x.shape[1] * m.scale_factor_0, x.shape[2] * m.scale_factor_0)
    x = self.conv(x, name + '.deconv2.0', act=ACT_RELU)
    x = self.conv(x, name + '.deconv2.2', act=ACT_RELU)
    x = self.upsample(x, x.shape[1] * m.scale_factor_1, x.shape[2] * m.scale_factor_1)
    x = self.conv(x, name + '.deconv3.0', act=ACT_RELU)
    return self.conv(x, name + '.deconv3.2')
