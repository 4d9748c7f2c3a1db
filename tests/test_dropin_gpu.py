"""The drop-in training step (carla_garage_amd/dropin.py) driven the way team_code/train.py:776-910 drives the reference's module:
forward (keyword arguments) -> model.compute_loss -> weighted sum with .item() on every loss -> backward -> optimizer.step ->
zero_grad(set_to_none=True); eager steps first, hipGraph replays after TFPP_DROPIN_GRAPH_AFTER of them.  Compared with the Trainer path
(pinned against the reference's own goldens in tests/test_model.py) on identical weights and batches.

/root/reference does not exist on the GPU box, so the reference's train.py itself cannot run here; tests/test_train_shim.py drives it on
the CPU of the build container up to the first forward, and `train_py_loop` below restates its inner loop line by line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.model import LidarCenterNet
from oracle import tfpp_port as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(dtype='fp32'):
  m = LidarCenterNet(GlobalConfig(tfpp_dtype=dtype))
  m.load_state_dict(P.make_state_dict(), strict=True)
  m.cuda().train()
  for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
      mod.p = 0.0
  m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0
  return m


def _batches(n, bs=2):
  out = []
  for i in range(n):
    b = {k: v.cuda() for k, v in P.make_labels(bs).items()}
    for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(bs)):
      b[k] = v.cuda()
    b['rgb'] = (b['rgb'] + 3.0 * i).clamp(0, 255)  # a different batch per step, same labels
    out.append(b)
  return out


def train_py_loop(model, optimizer, batches, weights, wrapper=None):
  """team_code/train.py:883-910 (Engine.train) + 688-820 (load_data_compute_loss), restated: returns the per-step total losses."""
  net = wrapper if wrapper is not None else model
  totals = []
  optimizer.zero_grad(set_to_none=False)                                         # train.py:881
  for b in batches:
    pred = net(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])  # :776-780
    lab = {k: v for k, v in b.items() if k.endswith('_label')}
    lab.setdefault('velocity_label', None)
    lab.setdefault('brake_target_label', None)
    losses = model.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3],
                                pred_bev_semantic=pred[4], pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8],
                                selected_path=pred[9], **lab)                     # :784-820
    loss = torch.zeros(1, dtype=torch.float32, device='cuda')
    for key, value in losses.items():
      loss += weights[key] * value                                               # :895
      float(weights[key] * float(value.item()))                                  # :896 (a host sync per loss)
    loss.backward()                                                              # :898
    optimizer.step()                                                             # :908
    optimizer.zero_grad(set_to_none=True)                                        # :910
    totals.append(float(loss.item()))                                            # :913
  return totals


def _params_by_name(model):
  """every trainable parameter in named_parameters() order (the flat arena is laid out by gradient completion order, which a Trainer and the
  drop-in step adopt at different steps: engine.arena_layout)"""
  return torch.cat([p.detach().float().reshape(-1) for _, p in model.named_parameters() if p.requires_grad]).clone()


def _trainer_reference(batches, lr):
  from carla_garage_amd.trainer import Trainer
  m = _model()
  tr = Trainer(m, lr=lr)
  p0 = _params_by_name(m)
  totals = [tr.total_loss(tr.train_step(b)) for b in batches]
  torch.cuda.synchronize()
  return totals, (p0, _params_by_name(m)), tr


def _check_params(flat_param, ref, steps, lr):
  """AdamW moves every parameter by ~lr per step whatever the size of its gradient: parameters whose gradient is pure rounding noise
  (structurally-zero key biases, pre-BN conv biases) take +-lr steps with a run-dependent sign, so element-wise equality cannot hold.
  The UPDATE (parameters minus their initial values) must agree in relative L2 and no element may be further off than opposite-sign steps."""
  p0, want = ref
  d_ref, d_got = (want - p0).double(), (flat_param - p0).double()
  rel = float((d_got - d_ref).norm() / d_ref.norm())
  worst = float((d_got - d_ref).abs().max())
  frac = float(((d_got - d_ref).abs() > 0.5 * lr).double().mean())
  assert rel < 0.1 and worst <= 2.2 * steps * lr and frac < 0.01, (rel, worst, frac)
  return rel, worst, frac


def test_train_py_loop_with_the_fused_optimizer_matches_the_trainer_eager_and_graph_replayed():
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  lr = 1e-4
  batches = _batches(5)
  want, want_param, tr = _trainer_reference(batches, lr)
  m = _model()
  opt = FlatAdamW(m.parameters(), lr=lr, amsgrad=True)
  got = train_py_loop(m, opt, batches, normalized_loss_weights(m.config))
  step = m.__dict__['_dropin_step']
  plan = next(iter(step.plans.values()))
  assert plan.count == 5 and plan.F is not None and plan.L is not None and plan.B1 is not None   # steps 4..5 were hipGraph replays (1-2 eager, 3 eager in the observed arena layout)
  np.testing.assert_allclose(got, want, rtol=2e-3)
  # same arena order on both sides; AdamW divides by sqrt(v): structurally-zero gradients turn rounding noise into +-lr steps, so the
  # parameters are compared to a fraction of the lr steps taken (as tests/test_model.py does), the losses above are the tight check
  _check_params(_params_by_name(m), want_param, 5, lr)
  # every .grad is None after zero_grad(set_to_none=True); the optimizer state has torch's layout
  assert all(p.grad is None for p in m.parameters())
  sd = opt.state_dict()
  assert len(sd['param_groups']) == 1 and sd['param_groups'][0]['amsgrad'] and len(sd['state']) == len([p for p in m.parameters() if p.requires_grad])
  assert float(sd['state'][min(sd['state'])]['step']) == 5.0   # (index 0 / 1 are the frozen visibility masks: no state, as in torch)


def test_train_py_loop_with_torch_adamw_matches_the_trainer():
  """The unmodified optimizer of train.py:529-531: .grad are views of the flat arena, torch.optim.AdamW updates the arena parameters in place."""
  from carla_garage_amd.losses import normalized_loss_weights
  lr = 1e-4
  batches = _batches(4)
  want, want_param, tr = _trainer_reference(batches, lr)
  m = _model()
  opt = torch.optim.AdamW(m.parameters(), lr=lr, amsgrad=True)
  got = train_py_loop(m, opt, batches, normalized_loss_weights(m.config))
  np.testing.assert_allclose(got, want, rtol=2e-3)
  step = m.__dict__['_dropin_step']
  _check_params(_params_by_name(m), want_param, 4, lr)


def test_optimizer_groups_fused_optimizer_matches_torch_adamw_with_the_same_groups():
  """use_optim_groups (train.py:522-523): ``create_optimizer_groups`` hands the optimizer a decay and a no-decay group.  FlatAdamW runs them as
  one launch with a bit per 4 arena elements; torch.optim.AdamW on the same groups (arena-backed parameters) is the reference.  A large weight
  decay makes a wrong group assignment visible; the state_dict has torch's two-group layout and survives a reload."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  lr, wd, steps = 1e-4, 5.0, 3   # (a weight decay large enough that a parameter in the wrong group moves by 1.5e-3 |w|, far above AdamW's +-lr noise)
  batches = _batches(steps)
  res = {}
  for kind in ('torch', 'fused'):
    m = _model()
    groups = m.create_optimizer_groups(wd)
    assert [g['weight_decay'] for g in groups] == [wd, 0.0]
    opt = (torch.optim.AdamW if kind == 'torch' else FlatAdamW)(groups, lr=lr, amsgrad=True)
    p0 = {n: p.detach().clone() for n, p in m.named_parameters()}
    losses = train_py_loop(m, opt, batches, normalized_loss_weights(m.config))
    res[kind] = (m, opt, p0, losses)
  (mt, _, p0, lt), (mf, of, _, lf) = res['torch'], res['fused']
  np.testing.assert_allclose(lf, lt, rtol=2e-3)
  worst = 0.0
  for (n, pt), (_, pf) in zip(mt.named_parameters(), mf.named_parameters()):
    if not pt.requires_grad:
      continue
    w0 = p0[n].double()
    assert float((pt - pf).detach().abs().max()) <= 2.2 * steps * lr + 1e-6, n   # element-wise: opposite-sign AdamW steps at most
    if pt.numel() >= 64 and float(w0.abs().mean()) >= 0.02:
      # relative shrink of the tensor, fused minus torch: 0 when both decay (or both do not) up to the sign-flip noise of AdamW on near-zero
      # gradients (measured worst 4.6e-4 over ALL tensors, tiny-weight ones included); +-lr wd steps = 1.5e-3 for a parameter in the wrong group
      r = float(((pf - pt).double() * torch.sign(w0)).sum() / w0.abs().sum())
      worst = max(worst, abs(r))
  assert worst < 7e-4, worst
  sd = of.state_dict()
  assert len(sd['param_groups']) == 2 and [g['weight_decay'] for g in sd['param_groups']] == [wd, 0.0]
  assert sum(len(g['params']) for g in sd['param_groups']) == len(list(mf.parameters()))
  assert sd['param_groups'][1]['params'][0] == len(sd['param_groups'][0]['params'])
  of2 = FlatAdamW(mf.create_optimizer_groups(wd), lr=lr, amsgrad=True)
  of2.load_state_dict(sd)
  sd2 = of2.state_dict()
  k = min(sd['state'])
  assert torch.equal(sd2['state'][k]['exp_avg'], sd['state'][k]['exp_avg']) and float(sd2['state'][k]['step']) == float(steps)


def test_gradients_are_arena_views_accumulate_and_match_the_engine():
  from carla_garage_amd.losses import normalized_loss_weights
  m = _model()
  b = _batches(1)[0]
  w = normalized_loss_weights(m.config)

  def fwd_bwd():
    pred = m(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])
    lab = {k: v for k, v in b.items() if k.endswith('_label')}
    losses = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                            pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
    total = sum(w[k] * v for k, v in losses.items())
    total.backward()
    return pred, losses, total

  pred, losses, total = fwd_bwd()
  step = m.__dict__['_dropin_step']
  eng = step.eng
  anchor = step.anchor
  for n, p in m.named_parameters():
    if p.requires_grad and p is not anchor:
      assert p.grad.data_ptr() == eng.grads[n].data_ptr(), n   # no copy: .grad IS the arena slice
  g1 = eng.flat_grad.clone()
  a1 = anchor.grad.clone()
  assert float(g1.abs().sum()) > 0 and float(a1.abs().sum()) > 0
  fwd_bwd()                                                     # second backward without zero_grad: gradients accumulate
  torch.cuda.synchronize()
  slot = eng.g(anchor)
  keep = torch.ones_like(g1, dtype=torch.bool)
  off = slot.data_ptr() - eng.flat_grad.data_ptr()
  keep[off // 4: off // 4 + slot.numel()] = False               # (the anchor's slot holds the last backward only; autograd accumulates anchor.grad)
  err = float(((eng.flat_grad - 2 * g1)[keep]).abs().max() / g1.abs().max())
  assert err < 1e-5, err
  for n, g in eng.grads.items():  # ... and per tensor (a kernel that overwrites instead of adding shows up as a factor 2 on ITS tensor)
    a = g1[(g.data_ptr() - eng.flat_grad.data_ptr()) // 4:][:g.numel()].view(g.shape)
    if eng.g(anchor).data_ptr() != g.data_ptr() and float(a.abs().max()) > 1e-4 * float(g1.abs().max()):
      assert float((g - 2 * a).abs().max()) <= 1e-3 * float(a.abs().max()), n
  assert float((anchor.grad - 2 * a1).abs().max()) <= 1e-6 * float(a1.abs().max()) + 1e-12
  # compute_loss on predictions that are NOT this forward's outputs (clones) takes the general path and gives the same values
  m.zero_grad(set_to_none=True)
  pred, losses, total = fwd_bwd()
  lab = {k: v for k, v in b.items() if k.endswith('_label')}
  with torch.no_grad():
    again = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1].clone(), pred_checkpoint=pred[2].clone(), pred_semantic=pred[3].clone(),
                           pred_bev_semantic=pred[4].clone(), pred_depth=pred[5].clone(), pred_bounding_box=tuple(x.clone() if x is not None else None for x in pred[6]),
                           pred_wp_1=pred[8], selected_path=pred[9], **lab)
  for k in losses:
    np.testing.assert_allclose(float(again[k]), float(losses[k]), rtol=2e-5, err_msg=k)


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def test_under_distributed_data_parallel_one_rank_rccl():
  """DistributedDataParallel(model) exactly as train.py:516-520 wraps it, 1-rank RCCL group, TFPP_FORCE_COLLECTIVES=1: DDP manages the anchor
  parameter only, the arena is averaged by one all-reduce per gradient bucket (each behind its completion signal), eager and replayed."""
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dropin_ddp_worker.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                     timeout=900, check=False)
  text = p.stdout.decode()
  assert p.returncode == 0, text[-4000:]
  r = json.loads([l for l in text.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):])
  assert r['ddp_params'] == 1
  k = r['buckets']
  assert k >= 4 and r['calls_per_step'] == [2, 2, k, k, k]  # the two static buckets, then the observed ones (DDP's own bucket of the anchor goes through its C++ reducer)
  assert r['arena_bytes_per_step'] == r['arena_bytes_at_step']   # together exactly one pass over the gradient arena per step
  assert r['graph_steps'] >= 2 and r['early_signals'] == k - 1 and r['wait_timeouts'] == 0, r
  assert max(r['loss_rel']) < 2e-3, r
  assert r['param_rel'] < 0.1 and r['param_abs'] <= 2.2 * 4 * r['lr'], r
