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


class Trainer:

  def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, process_group=None, use_graph=False):
    self.model = model
    self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
    self.pg = process_group
    self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
    self.step_count = 0
    self.use_graph = use_graph
    self.graph = None
    self._static = None
    self.eng = model._engine()
    self.loss_names = active_losses(model.config)
    self.loss_weights = normalized_loss_weights(model.config)
    self._flatten()
    self.seed_offset = ops.zeros(1, torch.int64, self.eng.device)  # advanced once per step (fresh dropout masks under replay)
    ops.set_seed_offset(self.seed_offset)

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
    self.exp_avg = ops.zeros(self.flat_param.shape, F32, dev)
    self.exp_avg_sq = ops.zeros(self.flat_param.shape, F32, dev)
    self.max_exp_avg_sq = ops.zeros(self.flat_param.shape, F32, dev)
    tdist.broadcast_state(self.flat_param, list(self.model.buffers()), self.pg)  # DDP constructor broadcast, train.py:516

  # ---------------------------------------------------------------------------------------------- one step
  def _step_part1(self, batch, split=True):
    """repack + forward + losses + the first backward segment (back to the end of fusion stage 3): on return the tail of the
    gradient arena, flat_grad[eng.early_offset:], is final.  split=False runs the whole backward in one segment (no join of the
    lanes in the middle: the single-GPU step)."""
    eng, model = self.eng, self.model
    eng.training = True
    eng.dtype = model.compute_dtype
    eng.repack(eng.dtype, True)
    eng.alloc_grads()
    ops.inc_u64(self.seed_offset)
    eng._seed_ctr = 0  # the per-call part of the seeds is a function of the call site only
    eng.tape = Tape(eng.lanes)
    t = eng.forward(batch['rgb'], batch['lidar_bev'], batch['target_point'], batch['ego_vel'], batch['command'])
    _, vals, seeds = fused_losses(model, t, batch, self.loss_weights, True)
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
    return vals

  def _optimizer(self, step):
    ops.adamw_amsgrad(self.flat_param, self.eng.flat_grad, self.exp_avg, self.exp_avg_sq, self.max_exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, step, grad_scale=1.0 / self.world)

  def train_step(self, batch):
    """batch: dict with rgb, lidar_bev, target_point, ego_vel, command and the *_label tensors (reference layouts).
    Returns the vector of unweighted losses (device tensor, order = self.loss_names)."""
    self.model.train()
    self.step_count += 1
    if not self.overlap_enabled():
      vals = self._step_body(batch)
      self.finish_step()
      return vals
    vals = self._step_part1(batch)
    early = self.reduce_early()  # N > 1: the finished two thirds of the gradients travel while the rest is computed
    self._step_part2()
    self.finish_step(early)
    return vals

  def overlap_enabled(self):
    return self.world > 1 or os.environ.get('TFPP_SPLIT_STEP', '0') == '1'

  def reduce_early(self):
    """Asynchronous all-reduce of flat_grad[early_offset:] (RCCL runs it on its own stream, ordered after what this stream has
    issued so far).  Returns the handle finish_step() waits on, or None when there is nothing to overlap."""
    if self.world == 1:
      return None
    return tdist.all_reduce_async(self.eng.flat_grad[self.eng.early_offset:], self.pg)

  def finish_step(self, early=None):
    """rest of the gradient exchange + optimizer (kept outside any captured graph)."""
    if early is None:
      tdist.all_reduce_gradients(self.eng.flat_grad, self.pg)
    else:
      tdist.all_reduce_gradients(self.eng.flat_grad[:self.eng.early_offset], self.pg)
      early.wait()
    self._optimizer(self.step_count)

  def total_loss(self, vals):
    w = torch.tensor([self.loss_weights[n] for n in self.loss_names])
    return float((vals.detach().cpu() * w).sum())
