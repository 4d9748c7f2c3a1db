"""Training step of the MI355X path: the data-parallel hot loop of team_code/train.py:883-910 without autograd.

One step = repack weights -> forward (engine tape) -> fused loss+gradient kernels -> tape backward into ONE flat fp32
gradient arena -> (RCCL all-reduce of that arena, the path's only collective: plain data parallelism,
team_code/train.py:516-520) -> fused AdamW(amsgrad) over the flat parameter arena (team_code/train.py:529-531,908).
The whole step is a static launch sequence, so after warm-up it is captured into ONE hipGraph (graph.GraphedTrainStep) and replayed
without Python or launch overhead -- with one rank and with eight: the arena is laid out in the order in which backward completes the
gradients (engine.arena_layout, observed on the first eager passes), every bucket is all-reduced behind its own completion event while
the rest of backward runs (buckets.py), and the optimizer updates a bucket as soon as it has landed.
"""

import torch
import torch.distributed as dist

from . import ops
from . import dist as tdist
from .engine import Tape, F32, arena_layout
from .losses import fused_losses, normalized_loss_weights, active_losses


class Trainer:

  def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, process_group=None, use_graph=False, lazy_state=False):
    """lazy_state: allocate the three optimizer-state arenas on the first optimizer launch (the drop-in path of dropin.py owns a Trainer
    for its flat arenas and gradient exchange; the 1.4 GB of AdamW state is only needed when the fused optimizer is the one stepping)."""
    self.model = model
    prev = model.__dict__.get('_trainer')
    if prev is not None and prev is not self:
      prev.detached = True  # the parameters move into THIS trainer's arena: the previous owner's arena is stale from here on
    model.__dict__['_trainer'] = self
    self.detached = False
    self._lazy_state = lazy_state
    self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay  # lr is read at every step: schedulers just set it
    self.pg = process_group
    self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    self.step_count = 0
    self.exchange = True  # False: skip the gradient collectives (bench.py times a local step beside the real one: exposed communication)
    self.use_graph = use_graph
    self.graph = None
    self._static = None
    self.eng = model._engine()
    self.loss_names = active_losses(model.config)
    self.loss_weights = normalized_loss_weights(model.config)
    self._flatten()
    # parameter groups (team_code/train.py:522-523): None = one group; else the ordered parameter lists of the groups and their weight decays --
    # the fused kernel supports ONE non-zero decay plus a weight_decay = 0 group (create_optimizer_groups, model.py:556-632) through a bit per
    # 4 arena elements
    self.groups, self.no_decay_bits = None, None
    if getattr(model.config, 'use_optim_groups', False) and not lazy_state:
      gr = model.create_optimizer_groups(weight_decay)
      self.set_groups([g['params'] for g in gr], [g['weight_decay'] for g in gr])
    self.seed_offset = ops.zeros(1, torch.int64, self.eng.device)  # advanced once per step (fresh dropout masks under replay)
    ops.set_seed_offset(self.seed_offset)
    self.layout_final = False  # the arenas are in the completion order of an observed backward pass (apply_observed_layout)
    self.eager_steps_in_layout = 0  # eager steps since the arenas last moved (a capture needs one: weight-image plans, scratch sizes)

  # ---------------------------------------------------------------------------------------------- flat arenas
  def _flatten(self):
    """Re-home every trainable parameter into one flat fp32 arena (same offsets as the gradient arena).  Called again when the bucket
    assignment changes (apply_observed_layout): values and optimizer state move with their parameters."""
    eng = self.eng
    old_state = None
    if getattr(self, 'exp_avg', None) is not None:  # (re-homing after optimizer steps: the moments follow their parameters)
      old_state = (dict(self._offsets), self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq)
    eng.alloc_grads()
    layout, total, _ = arena_layout(self.model)  # same layout as the gradient arena
    new_param = torch.empty(total, device=eng.device, dtype=F32)
    ops.zero_(new_param)
    for _, p, off in layout:
      n = p.numel()
      dst = new_param[off:off + n]
      ops.copy_rows(p.detach().contiguous(), dst, 1, n, 0, 0, 0, 0)
      p.data = dst.view(p.shape)
      p._tfpp_arena = (self, off)  # lets carla_garage_amd.optim.FlatAdamW find the arena a parameter lives in
    self.flat_param = new_param
    self._offsets = {id(p): off for _, p, off in layout}  # where every parameter lives in THIS layout (the next re-homing reads it)
    self._slices = None
    self.exp_avg = self.exp_avg_sq = self.max_exp_avg_sq = None
    if old_state is not None:
      old_off, *old_arenas = old_state
      self._alloc_state()
      for src, dst in zip(old_arenas, (self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq)):
        for _, p, off in layout:
          n = p.numel()
          ops.copy_rows(src[old_off[id(p)]:old_off[id(p)] + n], dst[off:off + n], 1, n, 0, 0, 0, 0)
    elif not self._lazy_state:
      self._alloc_state()
    if getattr(self, 'groups', None) is not None and getattr(self, 'no_decay_bits', None) is not None:
      self._groups_key = None  # the no-decay bit mask is a function of the offsets
      self.set_groups(self.groups, self.group_decays)
    eng.invalidate()
    if getattr(self, '_broadcast_done', False) is False:
      tdist.broadcast_state(self.flat_param, list(self.model.buffers()), self.pg)  # DDP constructor broadcast, train.py:516
      self._broadcast_done = True

  def apply_observed_layout(self):
    """Lay the arenas out in the completion order the last backward pass showed (Engine.end_backward), once that pass ran with the lane's
    final fork points (the forks sit at fractions of the PREVIOUS pass's closure count, so the second eager pass is the first valid one).
    Returns True when the arenas moved.  Never called while a captured graph holds pointers into them: graph.GraphedTrainStep and
    dropin.DropinStep run it on their eager warm-up steps only."""
    eng = self.eng
    obs = eng.observed_buckets
    if self.layout_final:
      return False
    if self.exchange and self.world > 1:
      # Agreed by construction: every rank adopts RANK 0's observation -- one small object broadcast per eager warm-up step until
      # rank 0 has a stable one -- so all ranks re-home on the same step with the same layout; agree_on_layout() stays as an assertion.
      # (Each rank acting on its OWN observation could re-home on different steps: one rank would then issue the layout check's int64
      # all-reduce while another issues a gradient bucket's float all-reduce.)
      box = [obs if (obs is not None and getattr(eng, 'observation_stable', False)) else None]
      # (needed whether or not THIS step exchanges -- no_sync() / accumulation steps re-home too -- and from the group's own rank 0)
      dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
      obs = box[0]
      if obs is None:
        return False
    elif obs is None or not getattr(eng, 'observation_stable', False):
      return False
    assign, flushes = obs
    self.layout_final = True
    cur = self.model.__dict__.get('_grad_buckets')
    if cur == assign and self.model.__dict__.get('_grad_bucket_flushes') == flushes:
      return False
    self.model.__dict__['_grad_buckets'] = dict(assign)
    self.model.__dict__['_grad_bucket_flushes'] = list(flushes)
    self._flatten()
    self.eager_steps_in_layout = 0
    return True

  def _alloc_state(self):
    if self.exp_avg is None:
      dev = self.eng.device
      self.exp_avg = ops.zeros(self.flat_param.shape, F32, dev)
      self.exp_avg_sq = ops.zeros(self.flat_param.shape, F32, dev)
      self.max_exp_avg_sq = ops.zeros(self.flat_param.shape, F32, dev)

  def arena_intact(self, quick=False):
    """True while every trainable parameter still lives at its place in flat_param (model.cuda() / .to() / .half() after the flattening
    re-allocate the parameters; the owner then has to flatten again)."""
    if self.detached:
      return False
    base = self.flat_param.data_ptr()
    sl = self._slices_cached()
    probe = sl if not quick else [sl[0], sl[len(sl) // 2], sl[-1]]
    return all(p.data_ptr() == base + 4 * off for _, off, p in probe)

  def _slices_cached(self):
    if getattr(self, '_slices', None) is None:
      self._slices = self._arena_slices()
    return self._slices

  # ---------------------------------------------------------------------------------------------- one step
  def _step_body(self, batch):
    """repack + forward + losses + backward: the part of the step that is captured into the hipGraph.  The lanes are joined once, at the
    end; the completion events of the gradient buckets are recorded on the weight-gradient lane as the pass goes (Engine._batch_end)."""
    eng, model = self.eng, self.model
    eng.training = True
    eng.dtype = model.compute_dtype
    eng.invalidate()  # this step rewrites the BatchNorm running statistics (and the optimizer the parameters) through raw pointers
    if ops.STAMPS['on']:
      ops.STAMPS['n'] = 0
    ops.stamp('step lane0 START')
    eng.repack(eng.dtype, True, defer=True)
    ops.stamp('step lane0 weights repacked')
    eng.alloc_grads(zero='beside_forward')  # (the 481 MB memset runs on the repack stream beside the first layers: -0.05 ms, A/B x3)
    ops.inc_u64(self.seed_offset)
    eng._seed_ctr = 0  # the per-call part of the seeds is a function of the call site only
    eng.tape = Tape(eng.lanes)
    t = eng.forward(batch['rgb'], batch['lidar_bev'], batch['target_point'], batch['ego_vel'], batch['command'])
    ops.stamp('step lane0 FORWARD DONE')
    _, vals, seeds = fused_losses(model, t, batch, self.loss_weights, True)
    ops.stamp('step lane0 losses done')
    tape, eng.tape = eng.tape, None
    eng.begin_backward()
    tape.backward(seeds)
    eng.end_backward()
    self.program = eng.bucket_program  # the completion signals this pass raises (a graph that replays it raises the same ones)
    return vals

  def set_groups(self, param_lists, weight_decays):
    """The optimizer's parameter groups in order (lists of parameters, frozen ones included as torch numbers them) and their weight decays:
    at most one non-zero value.  One group (or equal decays) = the plain kernel."""
    wds = [float(w) for w in weight_decays]
    nz = sorted({w for w in wds if w != 0.0})
    if len(nz) > 1:
      raise NotImplementedError(f'fused AdamW: one non-zero weight decay plus a weight_decay = 0 group (got {sorted(set(wds))})')
    key = (tuple(tuple(id(p) for p in pl) for pl in param_lists), tuple(wds))
    if getattr(self, '_groups_key', None) == key:
      return
    self._groups_key = key
    if len(param_lists) == 1 or len(set(wds)) == 1:
      self.groups, self.no_decay_bits = (None if len(param_lists) == 1 else [list(pl) for pl in param_lists]), None
      self.group_decays = wds
      self.weight_decay = wds[0]
      return
    self.groups, self.group_decays = [list(pl) for pl in param_lists], wds
    self.weight_decay = nz[0] if nz else 0.0
    no_decay = {id(p) for pl, w in zip(param_lists, wds) if w == 0.0 for p in pl}
    words = ops.no_decay_bitmask([(off, p.numel(), id(p)) for _, off, p in self._slices_cached()], no_decay, int(self.flat_param.numel()))
    self.no_decay_bits = torch.from_numpy(words.copy()).to(self.eng.device)

  def _group_index(self):
    """{id(param): position in the optimizer's numbering} (torch numbers the parameters group after group); None = model.parameters() order."""
    if self.groups is None:
      return None
    out, i = {}, 0
    for pl in self.groups:
      for p in pl:
        out[id(p)] = i
        i += 1
    return out

  def _optimizer(self, step, grad_scale=None, upto=None, lo=0):
    """grad_scale: None = 1 / world (the arena holds the SUM over the ranks); the drop-in path passes 1.0 (already averaged).
    upto / lo: only the elements lo .. upto-1 of the arena (finish_step updates bucket after bucket as the all-reduces land)."""
    self._alloc_state()
    self.eng.invalidate()
    sl = slice(lo, upto)
    bits = self.no_decay_bits
    if bits is not None and lo:
      assert lo % 128 == 0  # one 32-bit word of the mask covers 128 elements (buckets start on such an offset: engine.BUCKET_ALIGN)
      bits = bits[lo // 128:]
    ops.adamw_amsgrad(self.flat_param[sl], self.eng.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.max_exp_avg_sq[sl], self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, step, grad_scale=1.0 / self.world if grad_scale is None else grad_scale,
                      no_decay_bits=bits)

  def train_step(self, batch):
    """batch: dict with rgb, lidar_bev, target_point, ego_vel, command and the *_label tensors (reference layouts).
    Returns the vector of unweighted losses (device tensor, order = self.loss_names)."""
    self.model.train()
    self.apply_observed_layout()  # (eager steps only: a captured step never comes through here)
    self.step_count += 1
    self.eng.buckets.begin_issue()  # (serial number of this pass for its completion signals: in front of the pass, never inside a capture)
    vals = self._step_body(batch)
    self.finish_step(self.program)
    self.eager_steps_in_layout += 1
    return vals

  def exchange_enabled(self):
    return self.exchange and tdist.exchange_enabled(self.pg)

  def agree_on_layout(self):
    """Every rank observes its own backward pass: before the first exchange in a new arena layout the ranks compare it (bucket boundaries,
    parameter order) and fail loudly on a mismatch instead of averaging different parameters (one tiny all-reduce per layout)."""
    offs = getattr(self, '_offsets', None)  # (None: no arena was flattened -- the schedule test drives finish_step on a stand-in)
    if offs is None or getattr(self, '_layout_agreed', None) is offs or not tdist.exchange_enabled(self.pg):
      return
    import zlib
    sig = repr((tuple(self.eng.buckets.offsets), tuple((n, off) for n, _, off in arena_layout(self.model)[0]))).encode()
    tdist.assert_same_on_every_rank(zlib.crc32(sig) | (len(sig) << 32), 'the layout of the gradient arena', self.eng.flat_grad.device, self.pg)
    self._layout_agreed = offs

  def finish_step(self, program):
    """Gradient exchange + optimizer (outside any captured graph: RCCL is never captured).  ``program``: the completion signals of the pass
    that was just issued (Trainer.program after an eager pass, GraphedTrainStep.program for a replay).  One asynchronous all-reduce per
    bucket of the arena, each ordered behind the signal that says "this bucket is complete" -- they start while backward (the graph replay
    in flight) is still running -- and one optimizer launch per bucket as soon as it has landed; without an exchange, one launch over the
    whole arena."""
    buckets = self.eng.buckets
    if self.exchange:
      buckets.raise_if_timed_out()  # a signal wait of an EARLIER step gave up (on any rank): stop before this step's gradients are averaged in
      self.agree_on_layout()
    works = buckets.exchange(self.eng.flat_grad, program, self.pg) if self.exchange else []
    if not works:
      self._optimizer(self.step_count)
      return
    for (lo, hi), work in zip(self.eng.buckets.ranges(), works):
      if work is None:
        continue
      work.wait()
      self._optimizer(self.step_count, lo=lo, upto=hi)

  # ---------------------------------------------------------------------------------------------- checkpoint / resume
  def _arena_slices(self):
    """[(index in model.parameters() order, arena offset, parameter)] of the trainable parameters."""
    index = {id(p): i for i, p in enumerate(self.model.parameters())}
    return [(index[id(p)], off, p) for _, p, off in arena_layout(self.model)[0]]

  def check_exchange_health(self):
    """Host-side check (one device read): a completion-signal wait of the gradient exchange that gave up (buckets.WAIT_TIMEOUT_MS) means a
    collective may have run on an incomplete bucket.  Every step already looks at the previous step's time-out word without synchronising
    (GradBuckets.raise_if_timed_out in finish_step / DropinStep._run_backward); this is the blocking variant for checkpoints."""
    self.eng.buckets.raise_if_timed_out(block=True)
    n = self.eng.buckets.timed_out() if self.eng.buckets.timeouts is not None else 0
    if n:
      raise RuntimeError(f'carla_garage_amd: {n} completion-signal wait(s) of the gradient exchange timed out: gradients of those steps may be incomplete')

  def state_dict(self):
    """The optimizer state in the layout ``torch.optim.AdamW(model.parameters(), lr, amsgrad=True).state_dict()`` has in the
    reference (team_code/train.py:529-531, saved as optimizer_%04d.pth at train.py:967-976): per-parameter ``step``, ``exp_avg``,
    ``exp_avg_sq``, ``max_exp_avg_sq`` keyed by the position in ``model.parameters()`` (frozen parameters have no state), one
    parameter group.  Values are clones cut out of the flat arenas."""
    self.check_exchange_health()
    n_params = len(list(self.model.parameters()))
    state = {}
    self._alloc_state()
    gidx = self._group_index()
    if self.step_count > 0:
      for i, off, p in self._arena_slices():
        n = p.numel()
        if gidx is not None:
          i = gidx[id(p)]
        state[i] = {'step': torch.tensor(float(self.step_count)),
                    'exp_avg': self.exp_avg[off:off + n].view(p.shape).clone(),
                    'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).clone(),
                    'max_exp_avg_sq': self.max_exp_avg_sq[off:off + n].view(p.shape).clone()}
    group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay, 'amsgrad': True,
             'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
             'params': list(range(n_params))}
    if gidx is not None:  # torch's layout for several groups: consecutive indices group after group
      groups, i = [], 0
      for pl, wd in zip(self.groups, self.group_decays):
        groups.append({**group, 'weight_decay': wd, 'params': list(range(i, i + len(pl)))})
        i += len(pl)
      return {'state': state, 'param_groups': groups}
    return {'state': state, 'param_groups': [group]}

  def load_state_dict(self, sd):
    """Resume from ``state_dict()`` or from an optimizer_%04d.pth the reference wrote (train.py:533-534)."""
    g = sd['param_groups'][0]
    n_groups = 1 if self.groups is None else len(self.groups)
    if len(sd['param_groups']) != n_groups or not g.get('amsgrad', False):
      raise ValueError(f'expected {n_groups} amsgrad AdamW parameter group(s) (team_code/train.py:522-531), got {len(sd["param_groups"])}')
    self.lr, self.betas, self.eps = float(g['lr']), tuple(g['betas']), float(g['eps'])
    if self.groups is None:
      self.weight_decay = float(g['weight_decay'])
    else:
      if [len(x['params']) for x in sd['param_groups']] != [len(pl) for pl in self.groups]:
        raise ValueError('parameter groups of the checkpoint do not match the optimizer\'s')
      self._groups_key = None
      self.set_groups(self.groups, [float(x['weight_decay']) for x in sd['param_groups']])
    self._alloc_state()
    steps = set()
    gidx = self._group_index()
    for i, off, p in self._arena_slices():
      if gidx is not None:
        i = gidx[id(p)]
      st = sd['state'].get(i)
      n = p.numel()
      for name, arena in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq), ('max_exp_avg_sq', self.max_exp_avg_sq)):
        dst = arena[off:off + n]
        if st is None:
          ops.zero_(dst)
        else:
          src = st[name].detach().to(dst.device, F32).contiguous()
          ops.copy_rows(src, dst, 1, n, 0, 0, 0, 0)
      if st is not None:
        steps.add(int(float(st['step'])))
    if len(steps) > 1:
      raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): the fused optimizer keeps one step counter')
    self.step_count = steps.pop() if steps else 0
    self.eng.invalidate()

  def total_loss(self, vals):
    w = torch.tensor([self.loss_weights[n] for n in self.loss_names])
    return float((vals.detach().cpu() * w).sum())
