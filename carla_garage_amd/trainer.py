"""Training step of the MI355X path: the data-parallel hot loop of team_code/train.py:883-910 without autograd.

One step = repack weights -> forward (engine tape) -> fused loss+gradient kernels -> tape backward into ONE flat fp32
gradient arena -> (RCCL all-reduce of that arena, the path's only collective: plain data parallelism,
team_code/train.py:516-520) -> fused AdamW(amsgrad) over the flat parameter arena (team_code/train.py:529-531,908).
The whole step is a static launch sequence, so after warm-up it can be captured into a hipGraph
(``capture_graph=True``) and replayed without Python or launch overhead.
"""
import os

import torch
import torch.distributed as dist

from . import ops
from . import dist as tdist
from .engine import Tape, F32, arena_order
from .losses import fused_losses, normalized_loss_weights, active_losses

_PIPELINED_FINISH = os.environ.get('TFPP_PIPELINED_FINISH', '1') != '0'  # N > 1: optimizer on the early slice while the late slice is all-reduced
_EARLY_OPT = os.environ.get('TFPP_EARLY_OPTIMIZER', '0') == '1'  # single GPU: update the early-finishing slice of the arena inside the step; measured SLOWER (25.57 vs 25.05 ms/step: 0.6 ms of HBM streaming beside the latency-bound main chain), off by default


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
    # optimizer scalars in device memory (tfpp_adamw_amsgrad_dev): lets the update of the early-finishing slice of the arena run INSIDE the step
    self.hyper = ops.zeros(8, F32, self.eng.device)
    self._hyper_host = [torch.empty(8, dtype=F32).pin_memory() for _ in range(4)] if self.eng.device.type == 'cuda' else None
    self._hyper_i = 0
    self.early_opt_in_step = False  # set by the hook when the in-step launch was issued (or captured): finish_step then updates the rest only

  # ---------------------------------------------------------------------------------------------- flat arenas
  def _flatten(self):
    """Re-home every trainable parameter into one flat fp32 arena (same offsets as the gradient arena)."""
    eng = self.eng
    eng.alloc_grads()
    dev = eng.device
    params, _ = arena_order(self.model)  # same order as the gradient arena: late-finishing gradients first
    self.flat_param = torch.empty_like(eng.flat_grad)
    ops.zero_(self.flat_param)
    off = 0
    for _, p in params:
      n = p.numel()
      dst = self.flat_param[off:off + n]
      ops.copy_rows(p.detach().contiguous(), dst, 1, n, 0, 0, 0, 0)
      p.data = dst.view(p.shape)
      off += ops.pad_to(n, 4)
      p._tfpp_arena = (self, off - ops.pad_to(n, 4))  # lets carla_garage_amd.optim.FlatAdamW find the arena a parameter lives in
    self.exp_avg = self.exp_avg_sq = self.max_exp_avg_sq = None
    if not self._lazy_state:
      self._alloc_state()
    tdist.broadcast_state(self.flat_param, list(self.model.buffers()), self.pg)  # DDP constructor broadcast, train.py:516

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
  def _step_part1(self, batch, split=True):
    """repack + forward + losses + the first backward segment (back to the end of fusion stage 3): on return the tail of the
    gradient arena, flat_grad[eng.early_offset:], is final.  split=False runs the whole backward in one segment (no join of the
    lanes in the middle: the single-GPU step)."""
    eng, model = self.eng, self.model
    eng.training = True
    eng.dtype = model.compute_dtype
    eng.invalidate()  # this step rewrites the BatchNorm running statistics (and the optimizer the parameters) through raw pointers
    if ops.STAMPS['on']:
      ops.STAMPS['n'] = 0
    ops.stamp('step lane0 START')
    eng.repack(eng.dtype, True, defer=True)
    ops.stamp('step lane0 weights repacked')
    eng.alloc_grads()
    ops.inc_u64(self.seed_offset)
    eng._seed_ctr = 0  # the per-call part of the seeds is a function of the call site only
    eng.tape = Tape(eng.lanes)
    self.early_opt_in_step = False
    if _EARLY_OPT and not split and not self.overlap_enabled() and eng.side.enabled and self._hyper_host is not None and self.no_decay_bits is None:
      # single GPU: nothing has to be exchanged first, so the two thirds of the parameters whose gradients are final once backward has passed
      # Tape.mark() are updated on the weight-gradient lane while the stages 3..1 are still being differentiated
      eng.tape.on_mark = self._arm_early_optimizer
    t = eng.forward(batch['rgb'], batch['lidar_bev'], batch['target_point'], batch['ego_vel'], batch['command'])
    ops.stamp('step lane0 FORWARD DONE')
    _, vals, seeds = fused_losses(model, t, batch, self.loss_weights, True)
    ops.stamp('step lane0 losses done')
    self._tape, eng.tape = eng.tape, None
    self._tape.backward(seeds, stop_at_mark=split)
    return vals

  def _step_part2(self):
    """the rest of backward (stems .. fusion stage 3)"""
    self._tape.backward_resume()
    self._tape = None

  def _step_body(self, batch):
    vals = self._step_part1(batch, split=False)
    self._tape = None
    self.eng.side.after_mark, self.eng.side.mark_passed = None, False
    return vals

  def _arm_early_optimizer(self):
    side = self.eng.side
    side.mark_passed, side.after_mark = True, self._early_optimizer

  def _early_optimizer(self):
    off = self.eng.early_offset
    self._alloc_state()
    ops.adamw_amsgrad_dev(self.flat_param[off:], self.eng.flat_grad[off:], self.exp_avg[off:], self.exp_avg_sq[off:], self.max_exp_avg_sq[off:], self.hyper)
    self.early_opt_in_step = True

  def upload_hyper(self, step):
    """The optimizer scalars of update number ``step`` -> device (asynchronous copy from a pinned buffer; before the step that contains the launch)."""
    if self._hyper_host is None:
      return
    h = self._hyper_host[self._hyper_i % len(self._hyper_host)]
    self._hyper_i += 1
    bc1, bc2s = ops.adamw_bias_corrections(self.betas[0], self.betas[1], step)
    for i, v in enumerate((self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, 1.0 / self.world, bc1, bc2s)):
      h[i] = v
    self.hyper.copy_(h, non_blocking=True)

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
    upto / lo: only the elements lo .. upto-1 of the arena (the early-finishing slice is updated on its own: inside the step by
    _early_optimizer, or by finish_step while the late slice is still being all-reduced)."""
    self._alloc_state()
    self.eng.invalidate()
    sl = slice(lo, upto)
    bits = self.no_decay_bits
    if bits is not None and lo:
      assert lo % 128 == 0  # one 32-bit word of the mask covers 128 elements (finish_step only splits at such an offset)
      bits = bits[lo // 128:]
    ops.adamw_amsgrad(self.flat_param[sl], self.eng.flat_grad[sl], self.exp_avg[sl], self.exp_avg_sq[sl], self.max_exp_avg_sq[sl], self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, step, grad_scale=1.0 / self.world if grad_scale is None else grad_scale,
                      no_decay_bits=bits)

  def train_step(self, batch):
    """batch: dict with rgb, lidar_bev, target_point, ego_vel, command and the *_label tensors (reference layouts).
    Returns the vector of unweighted losses (device tensor, order = self.loss_names)."""
    self.model.train()
    self.step_count += 1
    if not self.overlap_enabled():
      self.upload_hyper(self.step_count)
      vals = self._step_body(batch)
      self.finish_step()
      return vals
    vals = self._step_part1(batch)
    early = self.reduce_early()  # N > 1: the finished two thirds of the gradients travel while the rest is computed
    self._step_part2()
    self.finish_step(early)
    return vals

  def overlap_enabled(self):
    return (self.exchange and tdist.exchange_enabled(self.pg)) or os.environ.get('TFPP_SPLIT_STEP', '0') == '1'

  def reduce_early(self, avg=False):
    """Asynchronous all-reduce of flat_grad[early_offset:] (RCCL runs it on its own stream, ordered after what this stream has
    issued so far).  Returns the handle finish_step() waits on, or None when there is nothing to overlap."""
    if not self.exchange:
      return None
    return tdist.all_reduce_async(self.eng.flat_grad[self.eng.early_offset:], self.pg, avg=avg)

  def finish_step(self, early=None):
    """rest of the gradient exchange + optimizer (kept outside any captured graph)."""
    if not self.exchange:
      pass
    elif early is None:
      tdist.all_reduce_gradients(self.eng.flat_grad, self.pg)
    else:
      off = self.eng.early_offset
      late = tdist.all_reduce_async(self.eng.flat_grad[:off], self.pg) if _PIPELINED_FINISH else None
      if late is None:
        tdist.all_reduce_gradients(self.eng.flat_grad[:off], self.pg)
      early.wait()
      if late is not None:
        # the late third of the arena (stems .. fusion stage 3) is the only part of the exchange nothing computes beside: the optimizer
        # updates the early two thirds (already reduced) while it travels -- 0.58 ms of the 0.87 ms launch hide 0.26 (direct) .. 1.8 ms (ring)
        split = not self.early_opt_in_step and (self.no_decay_bits is None or off % 128 == 0)
        if split:
          self._optimizer(self.step_count, lo=off)
        late.wait()
        self._optimizer(self.step_count, upto=off if (split or self.early_opt_in_step) else None)
        return
    self._optimizer(self.step_count, upto=self.eng.early_offset if self.early_opt_in_step else None)

  # ---------------------------------------------------------------------------------------------- checkpoint / resume
  def _arena_slices(self):
    """[(index in model.parameters() order, arena offset, parameter)] of the trainable parameters."""
    index = {id(p): i for i, p in enumerate(self.model.parameters())}
    out, off = [], 0
    for _, p in arena_order(self.model)[0]:
      out.append((index[id(p)], off, p))
      off += ops.pad_to(p.numel(), 4)
    return out

  def state_dict(self):
    """The optimizer state in the layout ``torch.optim.AdamW(model.parameters(), lr, amsgrad=True).state_dict()`` has in the
    reference (team_code/train.py:529-531, saved as optimizer_%04d.pth at train.py:967-976): per-parameter ``step``, ``exp_avg``,
    ``exp_avg_sq``, ``max_exp_avg_sq`` keyed by the position in ``model.parameters()`` (frozen parameters have no state), one
    parameter group.  Values are clones cut out of the flat arenas."""
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
