"""The training step behind the drop-in boundary, driven the way team_code/train.py drives the reference's module:

    pred = model(rgb=..., lidar_bev=..., ...)              # train.py:776-780   (DistributedDataParallel.forward -> LidarCenterNet.forward)
    losses = model.module.compute_loss(pred..., labels...)  # train.py:784-820
    loss = sum(weight[k] * losses[k]); loss.backward()      # train.py:887-898
    optimizer.step(); optimizer.zero_grad(set_to_none=True)  # train.py:908-910

Round 2 ran this at 96 ms/step against 30 ms for ``Trainer``: ~3000 launches issued from Python per step, every prediction gradient
converted NHWC -> NCHW -> NHWC on its way through autograd, 1332 parameter gradients cloned for autograd and copied twice more by DDP's
reducer.  ``DropinStep`` keeps the call sites and removes the overheads:

* parameters live in ONE flat fp32 arena, gradients in one flat arena (the ``Trainer`` of trainer.py, optimizer state allocated lazily);
  ``p.grad`` is set to cached views of the gradient arena (no copy), DDP only manages one small anchor parameter
  (``LidarCenterNet._ddp_params_and_buffers_to_ignore``) and the arena is all-reduced here bucket by bucket behind the buckets' completion signals (buckets.py), while backward still runs;
* ``compute_loss`` returns scalars whose backward hands autograd a zero-stride token instead of a full-resolution tensor; the gradient
  of each loss w.r.t. the internal NHWC prediction is already in HBM (fused loss kernels) and is only scaled by the incoming d(total)/d(loss);
* after ``TFPP_DROPIN_GRAPH_AFTER`` (2) eager steps of one input signature the three phases (forward / losses / backward) are captured
  into hipGraphs that share one memory pool and are replayed from then on; the predictions returned in that mode are STATIC tensors
  that the next step overwrites (train.py consumes them inside the step).

No fallback computes anything outside libtfpp_hip.so: unsupported uses raise."""
import os

import torch

from . import dist as tdist
from . import ops
from .engine import F32, Tape
from .graph import capture, capture_stream
from .losses import active_losses, fused_losses, output_slots

GRAPH_AFTER = int(os.environ.get('TFPP_DROPIN_GRAPH_AFTER', '2'))  # eager steps of one input signature before it is captured; < 0: never


class _Plan:
  """Everything that belongs to one input signature: call count and, once captured, the graphs and their static tensors."""

  def __init__(self):
    self.count = 0
    self.pool = None
    self.F = self.L = self.B1 = None
    self.program = None       # completion signals of the gradient buckets the captured backward raises (buckets.py)
    self.static_in = self.static_labels = None
    self.fwd = None           # dict(internal, tape, outs, diff) of the captured forward
    self.vals = self.loss_seeds = None
    self.broken = False       # a capture failed or the call sequence left the supported pattern: this signature stays eager


class _Step(torch.autograd.Function):
  """forward: the whole network (eager launches or one graph replay); backward: the whole backward pass into the flat gradient arena.
  The only differentiable input is the anchor parameter (see DropinStep.anchor)."""

  @staticmethod
  def forward(ctx, step, anchor, *inputs):
    outs, diff = step._run_forward(inputs)
    ctx.step, ctx.step_id = step, step.step_id
    ctx.set_materialize_grads(False)
    ctx.mark_non_differentiable(*[o for o, d in zip(outs, diff) if not d])
    return tuple(outs)

  @staticmethod
  def backward(ctx, *gouts):
    ganchor = ctx.step._run_backward(ctx.step_id, gouts)
    return (None, ganchor) + (None,) * 5


class _LossNode(torch.autograd.Function):
  """One loss scalar of compute_loss.  backward records d(total)/d(this loss) in DropinStep.gscale (a device scalar, no host sync)
  and hands autograd a zero-stride token of the prediction's shape: the real gradient never takes the caller-facing layout."""

  @staticmethod
  def forward(ctx, pred_caller, value, step, index, step_id):
    ctx.step, ctx.index, ctx.step_id = step, index, step_id
    ctx.shape, ctx.dtype, ctx.device = pred_caller.shape, pred_caller.dtype, pred_caller.device
    return value.detach()

  @staticmethod
  def backward(ctx, g):
    step = ctx.step
    if g is not None and ctx.step_id == step.step_id:
      ops.copy_rows(g.detach().float().reshape(1).contiguous(), step.gscale, 1, 1, 0, 0, 0, ctx.index)
      step.cur['used'].add(ctx.index)
    return step.token(ctx.shape, ctx.dtype, ctx.device), None, None, None, None


class _PairLossNode(torch.autograd.Function):
  """_LossNode of a loss with two caller-facing predictions (loss_wp of the multi_wp_output variant: pred_wp and pred_wp_1)."""

  @staticmethod
  def forward(ctx, caller0, caller1, value, step, index, step_id):
    ctx.step, ctx.index, ctx.step_id = step, index, step_id
    ctx.meta = [(c.shape, c.dtype, c.device) for c in (caller0, caller1)]
    return value.detach()

  @staticmethod
  def backward(ctx, g):
    step = ctx.step
    if g is not None and ctx.step_id == step.step_id:
      ops.copy_rows(g.detach().float().reshape(1).contiguous(), step.gscale, 1, 1, 0, 0, 0, ctx.index)
      step.cur['used'].add(ctx.index)
    return tuple(step.token(*m) for m in ctx.meta) + (None,) * 4


class DropinStep:

  def __init__(self, model):
    from .trainer import Trainer
    self.model = model
    self.cfg = model.config
    tr = model.__dict__.get('_trainer')
    if tr is None or not tr.arena_intact():
      tr = Trainer(model, lazy_state=True)
    self.tr, self.eng = tr, tr.eng
    self.names = active_losses(self.cfg)
    # the caller-facing predictions of model._export, each with the loss it belongs to (one per loss, except the two hypotheses of multi_wp_output)
    self.slots = output_slots(self.cfg)
    self.slot_loss = [self.names.index(l) for _, l in self.slots]
    self.gscale = ops.zeros(max(16, len(self.names)), F32, self.eng.device)
    self._tokens = {}
    self._views = None
    self.plans = {}
    self.cur = None
    self.step_id = 0

  # ------------------------------------------------------------------------------------------------ small helpers
  @property
  def anchor(self):
    return self.model._dropin_anchor()

  def token(self, shape, dtype, device):
    k = (dtype, str(device))
    t = self._tokens.get(k)
    if t is None:
      t = self._tokens[k] = ops.zeros(1, dtype, device).view(())
    return t.expand(shape)

  def _is_token(self, g):
    t = self._tokens.get((g.dtype, str(g.device)))
    return t is not None and g.data_ptr() == t.data_ptr() and all(s == 0 for s in g.stride())

  def _key(self, inputs):
    return (self.model.training, self.model.compute_dtype) + tuple((tuple(x.shape), x.dtype) for x in inputs)

  def _exchange_on(self):
    """Gradients are averaged over the ranks here when the module sits inside DistributedDataParallel (its constructor read
    _ddp_params_and_buffers_to_ignore) and there is more than one rank (or TFPP_FORCE_COLLECTIVES=1 for the 1-rank RCCL path)."""
    return self.model.__dict__.get('_ddp_seen', False) and tdist.exchange_enabled(self.tr.pg)

  def _grad_views(self):
    """[(parameter, cached view of its slice of the gradient arena)] for every parameter whose gradient the engine writes: the module's own
    parameters except the anchor (autograd delivers that one)."""
    if self._views is None or self._views[0] is not self.eng.flat_grad:
      self.eng.alloc_grads(zero=False)
      anchor = self.anchor
      own = self.model.__dict__['_own_param_names']
      named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad and p is not anchor]
      self._views = (self.eng.flat_grad, [(p, self.eng.grads[n]) for n, p in named if n in own],
                     [(p, self.eng.grads[n]) for n, p in named if n not in own])
    return self._views[1]

  def _foreign_views(self):
    """[(parameter, its slot in the gradient arena)] for trainable parameters the caller registered on the module (train.py:479-482: the learnable
    loss weights): autograd -- and DistributedDataParallel, which manages them -- deliver their gradients in ``.grad``; FlatAdamW copies them
    into the slot before the fused update, torch.optim.AdamW reads ``.grad`` as always."""
    self._grad_views()
    return self._views[2]

  def _grad_state(self):
    """'fresh': every .grad is None (zero_grad(set_to_none=True), train.py:910) -> the arena is zeroed and .grad set to its views;
    'accumulate': every .grad IS the cached arena view (zero_grad(set_to_none=False) or a second backward) -> the arena keeps its content;
    'foreign': anything else -> arena zeroed, the new gradients are added to whatever the caller put into .grad."""
    views = self._grad_views()
    if all(p.grad is None for p, _ in views):
      return 'fresh'
    if all(p.grad is v for p, v in views):
      return 'accumulate'
    return 'foreign'

  # ------------------------------------------------------------------------------------------------ forward
  def forward(self, inputs):
    return list(_Step.apply(self, self.anchor, *inputs))

  def _fwd_body(self, inputs):
    eng, model = self.eng, self.model
    eng.training = model.training
    eng.dtype = model.compute_dtype
    eng.invalidate()  # optimizers (fused or torch's, in place) and BatchNorm write parameters / statistics behind tensor._version
    eng.repack(eng.dtype, True, defer=True)
    ops.clear_stats_rows(eng.device)
    ops.set_seed_offset(self.tr.seed_offset)
    ops.inc_u64(self.tr.seed_offset)
    eng._seed_ctr = 0
    eng.tape = Tape(eng.lanes)
    internal = eng.forward(*inputs)
    tape, eng.tape = eng.tape, None
    outs, seeds = model._export(internal)
    assert len(outs) == len(self.slots), 'the caller-facing predictions are those of losses.output_slots (model._export)'
    return dict(internal=internal, tape=tape, outs=outs, export_seeds=seeds)

  def _run_forward(self, inputs):
    key = self._key(inputs)
    plan = self.plans.setdefault(key, _Plan())
    plan.count += 1
    self.step_id += 1
    cur = self.cur = dict(step_id=self.step_id, plan=plan, mode='eager', fwd=None, loss=None, used=set())
    tr = self.tr
    if not any(pl.F is not None for pl in self.plans.values()):
      # nothing is captured yet: the arenas may still move into the completion order the previous backward showed (trainer.py).  Only while no
      # parameter holds a view of the old gradient arena (train.py:910 zero_grad(set_to_none=True)); after a few steps of a caller that keeps
      # its .grad tensors the static layout is final (the exchange then starts when backward ends).
      if not tr.layout_final and all(p.grad is None for p, _ in self._grad_views()):
        if tr.apply_observed_layout():
          self._views = None
      if not tr.layout_final and plan.count > GRAPH_AFTER + 4:
        tr.layout_final = True
    if plan.F is not None and not plan.broken:
      for dst, src in zip(plan.static_in, inputs):
        dst.copy_(src, non_blocking=True)
      plan.F.replay()
      cur['mode'], cur['fwd'] = 'graph', plan.fwd
    elif GRAPH_AFTER >= 0 and not self.eng.sync_bn and plan.count > GRAPH_AFTER and not plan.broken and plan.count - 1 > 0 and tr.layout_final and tr.eager_steps_in_layout >= 1:
      plan.static_in = [x.detach().clone() for x in inputs]
      torch.cuda.synchronize()
      plan.F = torch.cuda.CUDAGraph()
      st = capture_stream(self.eng.device)
      with capture(plan.F, st):
        plan.fwd = self._fwd_body(plan.static_in)
      plan.pool = plan.F.pool()
      plan.F.replay()
      cur['mode'], cur['fwd'] = 'graph', plan.fwd
    else:
      cur['fwd'] = self._fwd_body([x.detach() for x in inputs])
    fwd = cur['fwd']
    self.model.__dict__['_last_internal'] = fwd['internal']
    outs = [o.detach() for o in fwd['outs']]  # fresh tensor objects (aliases): autograd attaches this call's node to them
    self.model.__dict__['_last_output_ptrs'] = {o.data_ptr() for o in outs}
    cur['out_ptrs'] = [o.data_ptr() for o in outs]
    return outs, [s is not None for s in fwd['export_seeds']]

  # ------------------------------------------------------------------------------------------------ losses
  def owns(self, callers):
    """True when ``callers`` (loss name -> caller-facing prediction) are exactly the predictions of the forward this object ran last."""
    cur = self.cur
    if cur is None or cur['loss'] is not None or not torch.is_grad_enabled():
      return False
    got = [callers.get(k) for k, _ in self.slots]
    return all(c is not None and c.requires_grad and c.data_ptr() == p for c, p in zip(got, cur['out_ptrs']))

  def losses(self, callers, labels):
    """compute_loss on the predictions of the last forward: {name: 0-d tensor} connected to autograd through _LossNode."""
    cur, model = self.cur, self.model
    plan = cur['plan']
    if cur['mode'] == 'graph':
      if plan.L is None:
        plan.static_labels = {k: v.detach().clone() for k, v in labels.items()}
        torch.cuda.synchronize()
        plan.L = torch.cuda.CUDAGraph()
        st = capture_stream(self.eng.device)
        with capture(plan.L, st, pool=plan.pool):
          _, plan.vals, plan.loss_seeds = fused_losses(model, plan.fwd['internal'], plan.static_labels, None, True)
      else:
        if set(labels) != set(plan.static_labels):
          raise RuntimeError('compute_loss: the set of labels changed after the step was captured into a hipGraph')
        for k, dst in plan.static_labels.items():
          dst.copy_(labels[k], non_blocking=True)
      plan.L.replay()
      vals, seeds = plan.vals, plan.loss_seeds
    else:
      _, vals, seeds = fused_losses(model, cur['fwd']['internal'], labels, None, True)
    cur['loss'] = dict(vals=vals, seeds=seeds)
    ops.zero_(self.gscale)
    out = {}
    for i, n in enumerate(self.names):
      mine = [callers[k] for k, l in self.slots if l == n]
      node = _LossNode if len(mine) == 1 else _PairLossNode
      out[n] = node.apply(*mine, vals[i], self, i, cur['step_id'])
    return out

  # ------------------------------------------------------------------------------------------------ backward
  def _scaled_seeds(self, loss_seeds, which=None):
    from .losses import _scale_by_device_scalar
    return [(pred, _scale_by_device_scalar(dpred, self.gscale[i:i + 1])) for i, (pred, dpred) in enumerate(loss_seeds) if which is None or i in which]

  def _bwd(self, tape, seeds):
    eng = self.eng
    eng.alloc_grads(zero=False)
    eng.begin_backward()
    tape.backward(seeds)
    eng.end_backward()

  def _run_backward(self, step_id, gouts):
    cur = self.cur
    if cur is None or cur['step_id'] != step_id:
      raise RuntimeError('backward of a forward that is not the most recent one of this module: the MI355X path keeps the activations of ONE '
                         'forward (as team_code/train.py needs); call backward before the next forward')
    eng, tr = self.eng, self.tr
    plan, fwd = cur['plan'], cur['fwd']
    state = self._grad_state()
    if state != 'accumulate':
      ops.zero_(eng.flat_grad)
    ops.zero_(eng.g(self.anchor))  # the anchor's gradient travels through autograd (which accumulates it): its slot holds this backward only
    live = [j for j, g in enumerate(gouts) if g is not None]
    tokens = cur['loss'] is not None and live and all(self._is_token(gouts[j]) for j in live)
    if cur['mode'] == 'graph' and tokens and set(live) == set(range(len(self.slots))):
      if plan.B1 is None:
        torch.cuda.synchronize()
        st = capture_stream(eng.device)
        plan.B1 = torch.cuda.CUDAGraph()
        with capture(plan.B1, st, pool=plan.pool):
          self._bwd(fwd['tape'], self._scaled_seeds(plan.loss_seeds))
        plan.program = eng.bucket_program
      eng.buckets.begin_issue()  # serial number of this pass's completion signals (buckets.py): in front of the replay, outside the capture
      plan.B1.replay()
      program = plan.program
    elif cur['mode'] == 'graph':
      plan.broken = True
      raise RuntimeError('this step was captured into hipGraphs (forward -> compute_loss -> backward of every loss, team_code/train.py:776-898) '
                         'but backward arrived with gradients that do not come from compute_loss; the signature runs eagerly from the next '
                         'step on (TFPP_DROPIN_GRAPH_AFTER=-1 disables the capture altogether)')
    else:
      seeds = []
      if tokens:
        seeds = self._scaled_seeds(cur['loss']['seeds'], {self.slot_loss[j] for j in live})
      else:
        seeded = set()
        for j in live:
          g = gouts[j]
          if self._is_token(g) and cur['loss'] is not None:
            if self.slot_loss[j] not in seeded:  # (both hypotheses of multi_wp_output carry the token of the ONE loss_wp seed)
              seeds += self._scaled_seeds(cur['loss']['seeds'], {self.slot_loss[j]})
              seeded.add(self.slot_loss[j])
          else:
            seeds.append(fwd['export_seeds'][j](g.contiguous()))
      eng.buckets.begin_issue()
      self._bwd(fwd['tape'], seeds)
      program = eng.bucket_program
      tr.eager_steps_in_layout += 1
    if self._exchange_on():
      eng.buckets.raise_if_timed_out()  # a signal wait of an earlier step gave up (on any rank): raise here, not at a checkpoint the caller may never write
      # DistributedDataParallel semantics (train.py:516-520): .grad holds the MEAN over the ranks when backward returns.  One all-reduce per
      # bucket of the arena, each behind its own completion event (they started while the replay above was still running); the caller's
      # stream waits for all of them
      tr.agree_on_layout()
      for work in eng.buckets.exchange(eng.flat_grad, program, tr.pg, avg=True):
        if work is not None:
          work.wait()
    cur['fwd'] = None if cur['mode'] == 'eager' else fwd  # eager: the activations die with the tape
    views = self._grad_views()
    if state == 'fresh':
      for p, v in views:
        p.grad = v
    elif state == 'foreign':
      for p, v in views:
        if p.grad is None:
          p.grad = v
        elif p.grad is not v:
          p.grad.add_(v)
    tr.grads_fresh = True
    ga = eng.g(self.anchor)
    out = torch.empty_like(ga)
    ops.copy_rows(ga, out, 1, ga.numel(), 0, 0, 0, 0)
    return out
