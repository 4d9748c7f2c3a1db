"""``FlatAdamW``: torch.optim.Optimizer front of the fused AdamW(amsgrad) kernel over the flat parameter / gradient arenas.

team_code/train.py:527-531 builds ``optim.AdamW(params, lr=args.lr, amsgrad=True)`` (optionally inside ``ZeroRedundancyOptimizer``) and
calls ``optimizer.step()`` / ``zero_grad(set_to_none=True)`` / ``state_dict()`` / ``load_state_dict()`` (train.py:533-534,908-910,967-976) and
hands it to the LR schedulers (train.py:589-598).  This class answers the same calls; ``step()`` is ONE launch of ``tfpp_adamw_amsgrad`` per
model (4.3 GB of HBM traffic at ~5.8 TB/s) instead of torch's ~500 multi-tensor launches over 1332 tensors.  The integration is the
one-line substitution of INTEGRATION.md (``optim.AdamW`` -> ``carla_garage_amd.optim.FlatAdamW``); tools/reference_train_shim.py makes it
for the unmodified train.py.  ZeRO-1 sharding of the optimizer state is not reproduced (1.9 GB of state per rank against 288 GB of HBM).

The parameters must belong to carla_garage_amd.LidarCenterNet modules whose first training forward has run (that is when they move into
the flat arena, dropin.py); other parameters (train.py registers learnable loss weights on the module with ``--learn_multi_task_weights``)
are updated by the same kernel, one small launch each.  There is no ATen fallback."""
import torch

from . import ops
from .engine import F32


class FlatAdamW(torch.optim.Optimizer):

  def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=True, **unused):
    if not amsgrad:
      raise ValueError('FlatAdamW implements AdamW with amsgrad=True (team_code/train.py:529-531)')
    super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=True))
    self._loose = {}            # id(param) -> (m, v, vmax, steps) for parameters outside any arena
    self._updated = set()       # id(param) of arena parameters that have received a gradient at least once (the others carry no state, as in torch)
    self._pending_state = None  # load_state_dict before the arenas exist (train.py:533-534 resumes before the first forward)

  # ---------------------------------------------------------------------------------------------- arenas
  def _arenas(self):
    """{trainer: [(group, param)]} for the arena parameters, [(group, param)] for the rest."""
    arenas, loose = {}, []
    for group in self.param_groups:
      for p in group['params']:
        if not p.requires_grad:
          continue
        owner = getattr(p, '_tfpp_arena', None)
        if owner is not None and not owner[0].detached:
          arenas.setdefault(owner[0], []).append((group, p))
        else:
          loose.append((group, p))
    return arenas, loose

  @torch.no_grad()
  def step(self, closure=None):
    loss = None
    if closure is not None:
      with torch.enable_grad():
        loss = closure()
    plan = getattr(self, '_plan', None)
    if plan is None or any(tr.detached for tr in plan[0]):
      plan = self._plan = self._arenas()
      self._tr_wds = {}
      self._tr_groups = {}  # trainer -> the parameter groups that hold its parameters, in the optimizer's order (cached: 1332 lookups otherwise)
    arenas, loose = plan
    for tr, members in arenas.items():
      groups = self._tr_groups.get(tr)
      if groups is None:
        mine = {id(g) for g, _ in members}
        groups = self._tr_groups[tr] = [g for g in self.param_groups if id(g) in mine]
      group = groups[0]
      if len(groups) > 1:
        # use_optim_groups (train.py:522-523): the groups of create_optimizer_groups -- one hyper-parameter set, weight decay per group
        for g in groups[1:]:
          if (float(g['lr']), tuple(g['betas']), float(g['eps'])) != (float(group['lr']), tuple(group['betas']), float(group['eps'])):
            raise NotImplementedError('FlatAdamW: the parameter groups of one model share lr / betas / eps (they differ in weight_decay only)')
      wds = tuple(float(g['weight_decay']) for g in groups)
      if self._tr_wds.get(tr) != wds:  # (first step, or a scheduler / the caller changed a weight decay)
        tr.set_groups([g['params'] for g in groups], wds)
        self._tr_wds[tr] = wds
      if len(members) != len(tr._slices_cached()):
        raise ValueError('FlatAdamW: every trainable parameter of the model must be optimised by the same optimizer (the fused kernel updates the whole arena)')
      if self._pending_state is not None:
        tr.load_state_dict(self._pending_state)
        self._mark_loaded(tr, self._pending_state)
        self._pending_state = None
      # gradients that autograd / DDP delivered outside the arena (the anchor parameter, anything the caller assigned to .grad): copy them in
      step = tr.model.__dict__.get('_dropin_step')
      pairs = ([(p, tr.eng.g(p)) for _, p in members] if step is None or step.tr is not tr else
               step._grad_views() + step._foreign_views() + [(step.anchor, tr.eng.g(step.anchor))])  # (cached views: the identity test below is 0.2 ms for 1332 parameters)
      skipped = []
      for p, slot in pairs:
        g = p.grad
        if g is slot:
          self._updated.add(id(p))
          continue
        if g is None:
          # torch.optim.AdamW skips a parameter without a gradient (no decay, no moment update): the fused kernel walks the whole arena, so the
          # parameter and its three state slices are saved here and put back after the launch (unused heads; rare and small)
          ops.zero_(slot)
          skipped.append(p)
          continue
        self._updated.add(id(p))
        if g.data_ptr() != slot.data_ptr():
          ops.copy_rows(g.detach().float().contiguous(), slot, 1, slot.numel(), 0, 0, 0, 0)
      tr.lr, tr.betas, tr.eps = float(group['lr']), tuple(group['betas']), float(group['eps'])  # (the weight decay(s): set_groups above)
      saved = []
      if skipped:
        tr._alloc_state()
        base = tr.flat_param.data_ptr()
        for p in skipped:
          off, n = (p.data_ptr() - base) // 4, p.numel()
          saved.append((off, n, [a[off:off + n].clone() for a in (tr.flat_param, tr.exp_avg, tr.exp_avg_sq, tr.max_exp_avg_sq)]))
      tr.step_count += 1
      tr._optimizer(tr.step_count, grad_scale=1.0)
      for off, n, keep in saved:
        for a, k in zip((tr.flat_param, tr.exp_avg, tr.exp_avg_sq, tr.max_exp_avg_sq), keep):
          ops.copy_rows(k, a[off:off + n], 1, n, 0, 0, 0, 0)
    for group, p in loose:
      if p.grad is None:
        continue
      st = self._loose.get(id(p))
      if st is None:
        st = self._loose[id(p)] = [ops.zeros(p.numel(), F32, p.device), ops.zeros(p.numel(), F32, p.device), ops.zeros(p.numel(), F32, p.device), 0]
      st[3] += 1
      if p.dtype != F32 or not p.is_contiguous() or not p.is_cuda:
        raise NotImplementedError('FlatAdamW: parameters are fp32, contiguous, on the GPU')
      ops.adamw_amsgrad(p.data.view(-1), p.grad.detach().float().contiguous().view(-1), st[0], st[1], st[2], float(group['lr']), group['betas'][0],
                        group['betas'][1], float(group['eps']), float(group['weight_decay']), st[3], grad_scale=1.0)
    return loss

  def _mark_loaded(self, tr, state_dict):
    """Parameters that carry state in a loaded checkpoint count as updated (their state must survive the next state_dict())."""
    index = tr._group_index() or {id(p): i for i, p in enumerate(tr.model.parameters())}
    have = set(state_dict.get('state', {}).keys())
    self._updated |= {pid for pid, i in index.items() if i in have}

  # ---------------------------------------------------------------------------------------------- checkpoint / resume
  def state_dict(self):
    if self._pending_state is not None:  # loaded before the first step and not applied yet: that IS the state
      return self._pending_state
    arenas, loose = self._arenas()
    if len(arenas) == 1 and not loose:
      tr = next(iter(arenas))
      g = self.param_groups[0]
      tr.lr, tr.betas, tr.eps = float(g['lr']), tuple(g['betas']), float(g['eps'])
      tr.set_groups([x['params'] for x in self.param_groups], [x['weight_decay'] for x in self.param_groups])
      sd = tr.state_dict()  # torch.optim.AdamW's layout: positions in model.parameters() (one group) / group after group (trainer.py)
      # a parameter that never received a gradient (an unused head) has NO state in torch.optim.AdamW: drop the entry the arena-wide kernel
      # implies, so that a checkpoint loads into the reference's optimizer exactly as one written by it would
      if self._updated:
        index = tr._group_index() or {id(p): i for i, p in enumerate(tr.model.parameters())}  # (the numbering Trainer.state_dict used)
        never = {index[id(p)] for _, p in arenas[tr] if id(p) not in self._updated and id(p) in index}
        for i in never:
          sd['state'].pop(i, None)
      for x, out in zip(self.param_groups, sd['param_groups']):
        for k, v in x.items():  # keys the LR schedulers add to the group (initial_lr, ...)
          if k != 'params' and k not in out:
            out[k] = v
      return sd
    if not arenas:  # before the first training step: no state yet
      return {'state': {}, 'param_groups': [{**{k: v for k, v in g.items() if k != 'params'}, 'params': list(range(len(g['params'])))} for g in self.param_groups]}
    raise NotImplementedError('FlatAdamW.state_dict: the parameters of one model')

  def load_state_dict(self, state_dict):
    arenas, _ = self._arenas()
    if len(state_dict['param_groups']) != len(self.param_groups):
      raise ValueError(f"loaded state dict has {len(state_dict['param_groups'])} parameter groups, the optimizer {len(self.param_groups)}")
    for g, mine in zip(state_dict['param_groups'], self.param_groups):
      for k, v in g.items():
        if k != 'params':
          mine[k] = v
    if len(arenas) == 1:
      tr = next(iter(arenas))
      tr.set_groups([x['params'] for x in self.param_groups], [x['weight_decay'] for x in self.param_groups])
      tr.load_state_dict(state_dict)
      self._mark_loaded(tr, state_dict)
    else:
      self._pending_state = state_dict  # applied by the first step(), when the arenas exist
