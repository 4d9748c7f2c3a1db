"""Call sites of the drop-in boundary that team_code/train.py exercises by default or documents (VERDICT r3 missing #3), each driven the
way train.py drives it and pinned against the reference where the reference defines the answer:

* two-stage training, ``freeze_backbone`` (train.py:495-508): golden written by the unmodified reference (``make_golden freeze``);
* ``validate()`` (train.py:923-956): ``@torch.inference_mode()``, eval mode, forward + ``compute_loss``: golden (``make_golden validate``);
* ``use_grad_clip`` (train.py:900-906): ``clip_grad_norm_`` on the arena-backed ``.grad`` views, then the fused optimizer;
* ``learn_multi_task_weights`` (train.py:479-483,891-894): loss weights registered on the module as parameters, trained by the same optimizer;
* ``use_amp`` (train.py:599,885,898,903-909): ``torch.autocast`` + ``GradScaler`` around the step;
* ``ZeroRedundancyOptimizer`` (train.py:527-529, the reference's DEFAULT optimizer wrapper) and ``DistributedDataParallel`` with the learnable
  loss weights: one-rank RCCL group in a worker process (tests/boundary_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import parity_util as U
from oracle import tfpp_port as P
import test_model as TM
import test_dropin_gpu as TD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _freeze_like_train_py(m):
  m.backbone.requires_grad_(False)              # train.py:496
  m.head.requires_grad_(False)                  # :499
  m.semantic_decoder.requires_grad_(False)      # :502
  m.bev_semantic_decoder.requires_grad_(False)  # :505
  m.depth_decoder.requires_grad_(False)         # :508


def test_freeze_backbone_step_vs_reference_golden():
  """Stage two of the two-stage training: only the planning side is trainable.  Losses, the 130 per-parameter gradient norms / sampled
  elements and the BatchNorm running statistics (a frozen BatchNorm in train mode still updates them, as in the reference) against the
  reference's own step; the frozen parameters have no slot in the gradient arena at all."""
  m = TM._model()
  _freeze_like_train_py(m)
  g = U.load_golden('tfpp_train_freeze_bs2.npz')
  want = sorted(str(n) for n in g['grad_names'])
  _, eng = TM._check_train_step_vs_golden(2, 'tfpp_train_freeze_bs2.npz', 'train_fp32_freeze', model=m)
  assert sorted(eng.grads) == want and len(want) == 130
  assert not any(n.startswith(('backbone.', 'head.', 'semantic_decoder.', 'bev_semantic_decoder.', 'depth_decoder.')) for n in eng.grads)
  # ... and through the drop-in boundary with the reference's loop (train.py:883-910): the frozen parameters do not move
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  m2 = TD._model()
  _freeze_like_train_py(m2)
  frozen0 = {n: p.detach().clone() for n, p in m2.named_parameters() if not p.requires_grad}
  train0 = {n: p.detach().clone() for n, p in m2.named_parameters() if p.requires_grad}
  opt = FlatAdamW(m2.parameters(), lr=1e-3, amsgrad=True)  # train.py:529-531 passes every parameter, frozen ones included
  losses = TD.train_py_loop(m2, opt, TD._batches(3), normalized_loss_weights(m2.config))
  assert all(np.isfinite(losses))
  for n, p in m2.named_parameters():
    if n in frozen0:
      assert torch.equal(p.detach(), frozen0[n]), n
  moved = sum(int(not torch.equal(p.detach(), train0[n])) for n, p in m2.named_parameters() if n in train0)
  assert moved >= 0.9 * len(train0), (moved, len(train0))


def test_validate_inference_mode_eval_losses_vs_reference_golden():
  """Engine.validate of train.py: inference_mode + eval(): forward (running-statistic BatchNorm, no dropout), compute_loss on the returned
  predictions, `.item()` on every loss.  The ten losses at 1e-3 against the reference's."""
  g = U.load_golden('tfpp_validate_bs2.npz')
  m = TM._model()
  m.eval()
  inp = [x.cuda() for x in P.make_inputs(2)]
  lab = {k: v.cuda() for k, v in P.make_labels(2).items()}
  with torch.inference_mode():
    out = m(*inp)
    losses = m.compute_loss(pred_wp=out[0], pred_target_speed=out[1], pred_checkpoint=out[2], pred_semantic=out[3], pred_bev_semantic=out[4],
                            pred_depth=out[5], pred_bounding_box=out[6], pred_wp_1=out[8], selected_path=out[9], **lab)
    got = {k: float(v.item()) for k, v in losses.items()}
  want = dict(zip([str(x) for x in g['loss_names']], g['losses']))
  assert set(got) == set(want)
  errs = {k: abs(got[k] - want[k]) / abs(want[k]) for k in want}
  TM._report('validate_eval_losses', errs)
  assert max(errs.values()) <= 1e-3, errs
  # predictions kept across steps: the eval forward returns caller-owned tensors (nothing static is overwritten by the next call)
  keep = out[1].clone()
  with torch.inference_mode():
    m(*[x.cuda() for x in P.make_inputs(1)])
  assert torch.equal(out[1], keep)


def _loop(m, opt, batches, weights, clip=None, scaler=None, learn=None):
  """train.py:883-910 with its optional branches (grad clip, GradScaler + autocast, learnable loss weights)."""
  totals = []
  opt.zero_grad(set_to_none=False)
  for b in batches:
    with torch.autocast(device_type='cuda', dtype=torch.float16, enabled=scaler is not None and scaler.is_enabled()):  # train.py:885
      pred = m(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])
      lab = {k: v for k, v in b.items() if k.endswith('_label')}
      lab.setdefault('velocity_label', None)
      lab.setdefault('brake_target_label', None)
      losses = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                              pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
      loss = torch.zeros(1, dtype=torch.float32, device='cuda')
      for key, value in losses.items():
        if learn is not None:
          precision = torch.exp(-learn[key])            # train.py:892-893
          loss += precision * value + learn[key]
        else:
          loss += weights[key] * value
    (scaler.scale(loss) if scaler is not None else loss).backward()  # :898
    if clip is not None:
      if scaler is not None:
        scaler.unscale_(opt)                                        # :902
      torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=clip, error_if_nonfinite=True)  # :904-906
    if scaler is not None:
      scaler.step(opt)                                              # :908
      scaler.update()
    else:
      opt.step()
    opt.zero_grad(set_to_none=True)
    totals.append(float(loss.item()))
  return totals


def _params(m):
  return torch.cat([p.detach().float().reshape(-1) for _, p in m.named_parameters() if p.requires_grad]).clone()


def test_grad_clip_on_arena_views_then_fused_and_torch_optimizer_agree():
  """clip_grad_norm_ scales the `.grad` views in place -- that IS the gradient arena the fused optimizer reads.  The clipped step of
  FlatAdamW must equal the clipped step of torch.optim.AdamW on the same arena-backed parameters, and the clip must have been active."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  lr, clip = 1e-4, 1.0
  batches = TD._batches(3)
  res = {}
  for kind in ('torch', 'fused'):
    m = TD._model()
    opt = (torch.optim.AdamW if kind == 'torch' else FlatAdamW)(m.parameters(), lr=lr, amsgrad=True)
    p0 = _params(m)
    # the norm before clipping, once (no step): it has to exceed max_norm, otherwise the test exercises nothing
    if kind == 'torch':
      w = normalized_loss_weights(m.config)
      b = batches[0]
      pred = m(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])
      lab = {k: v for k, v in b.items() if k.endswith('_label')}
      ls = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                          pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
      sum(w[k] * v for k, v in ls.items()).backward()
      total = float(torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=clip, error_if_nonfinite=True))
      after = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None)))
      assert total > 2 * clip and abs(after - clip) < 1e-3 * clip, (total, after)
      step = m.__dict__['_dropin_step']
      assert abs(float(step.eng.flat_grad.double().norm()) - after) < 2e-3 * clip  # the views ARE the arena (anchor slot included via .grad)
      opt.zero_grad(set_to_none=True)
    res[kind] = (_loop(m, opt, batches, normalized_loss_weights(m.config), clip=clip), _params(m) - p0)
  np.testing.assert_allclose(res['fused'][0], res['torch'][0], rtol=2e-3)
  d_t, d_f = res['torch'][1].double(), res['fused'][1].double()
  rel = float((d_f - d_t).norm() / d_t.norm())
  assert rel < 0.1 and float((d_f - d_t).abs().max()) <= 2.2 * 3 * lr, rel


def test_learnable_loss_weights_registered_like_train_py():
  """--learn_multi_task_weights: train.py registers config.detailed_loss_weights[k] as `weight_<k>` parameters on the module BEFORE the DDP
  wrap and optimises them with everything else.  They are not the engine's: their gradients come from plain autograd
  (d/dw [exp(-w) L + w] = 1 - exp(-w) L), the optimizer (fused or torch's) updates them, and the model's own gradients carry exp(-w)."""
  from carla_garage_amd.optim import FlatAdamW
  lr = 1e-3
  batches = TD._batches(3)
  res = {}
  for kind in ('torch', 'fused'):
    m = TD._model()
    names = [n for n in ('loss_target_speed', 'loss_checkpoint', 'loss_semantic', 'loss_bev_semantic', 'loss_depth', 'loss_center_heatmap', 'loss_wh',
                         'loss_offset', 'loss_yaw_class', 'loss_yaw_res')]
    learn = {}
    for i, k in enumerate(names):
      w = torch.nn.Parameter(torch.tensor(0.1 * i, dtype=torch.float32))
      m.register_parameter(name='weight_' + k, param=w)          # train.py:481-482
      learn[k] = w
    m.cuda()                                                       # train.py:483
    opt = (torch.optim.AdamW if kind == 'torch' else FlatAdamW)(m.parameters(), lr=lr, amsgrad=True)
    assert not any(n.startswith('weight_') for n in m._ddp_params_and_buffers_to_ignore)  # DDP manages them
    w0 = {k: float(v) for k, v in learn.items()}
    totals = _loop(m, opt, batches, None, learn=learn)
    res[kind] = (totals, {k: float(v) for k, v in learn.items()}, w0)
  np.testing.assert_allclose(res['fused'][0], res['torch'][0], rtol=2e-3)
  for k in res['torch'][1]:
    dt_, df_ = res['torch'][1][k] - res['torch'][2][k], res['fused'][1][k] - res['fused'][2][k]
    assert abs(dt_) > 0.5 * lr, (k, dt_)                           # every weight moved (AdamW: ~lr per step)
    assert abs(df_ - dt_) <= 0.5 * lr + 0.05 * abs(dt_), (k, dt_, df_)


def test_amp_autocast_and_grad_scaler_leave_the_step_unchanged():
  """use_amp=1 wraps the step in torch.autocast(float16) and a GradScaler.  The HIP path computes in its own dtype whatever autocast says;
  the scaler multiplies the loss by a power of two, unscale_ divides the arena views by it: the parameters after three steps equal those of
  the plain loop (the scale is exact in fp32)."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  lr = 1e-4
  batches = TD._batches(3)
  out = {}
  for amp in (False, True):
    m = TD._model()
    opt = FlatAdamW(m.parameters(), lr=lr, amsgrad=True)
    p0 = _params(m)
    scaler = torch.amp.GradScaler('cuda', enabled=True, init_scale=1024.0) if amp else None
    out[amp] = (_loop(m, opt, batches, normalized_loss_weights(m.config), scaler=scaler, clip=1e9 if amp else None), _params(m) - p0)
  np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-4)
  d0, d1 = out[False][1].double(), out[True][1].double()
  assert float((d1 - d0).norm() / d0.norm()) < 0.05


def test_fused_optimizer_leaves_a_parameter_without_gradient_alone_like_torch_and_keeps_a_loaded_state():
  """torch.optim.AdamW skips a parameter whose .grad is None (no weight decay, no moment update); the fused kernel walks the whole arena, so
  FlatAdamW restores such a parameter and its state after the launch.  And a state loaded before the first step is what
  state_dict() returns until that step has applied it."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  m = TD._model()
  opt = FlatAdamW(m.parameters(), lr=1e-2, amsgrad=True, weight_decay=0.1)
  w = normalized_loss_weights(m.config)
  TD.train_py_loop(m, opt, TD._batches(1), w)                       # one ordinary step: arenas and state exist
  plist = list(m.parameters())
  victim, other = m.target_speed_network[2].bias, m.target_speed_network[0].weight
  vi = next(i for i, q in enumerate(plist) if q is victim)
  b = TD._batches(2)[1]
  pred = m(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])
  lab = {k: v for k, v in b.items() if k.endswith('_label')}
  ls = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                      pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
  sum(w[k] * v for k, v in ls.items()).backward()
  victim.grad = None
  before = (victim.detach().clone(), other.detach().clone(), {k: v.clone() for k, v in opt.state_dict()['state'][vi].items()})
  opt.step()
  torch.cuda.synchronize()
  assert torch.equal(victim.detach(), before[0])                    # untouched: not even decayed
  assert not torch.equal(other.detach(), before[1])                 # everything else stepped
  after = opt.state_dict()['state'][vi]
  for k in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq'):
    assert torch.equal(after[k], before[2][k]), k
  # a state loaded before the first step of a fresh optimizer is not lost by an early state_dict()
  sd = opt.state_dict()
  m2 = TD._model()
  opt2 = FlatAdamW(m2.parameters(), lr=1e-2, amsgrad=True, weight_decay=0.1)
  opt2.load_state_dict(sd)
  got = opt2.state_dict()
  assert got['state'].keys() == sd['state'].keys() and torch.equal(got['state'][vi]['exp_avg'], sd['state'][vi]['exp_avg'])
  TD.train_py_loop(m2, opt2, TD._batches(1), w)                      # the first step applies it
  assert float(opt2.state_dict()['state'][vi]['step']) == float(sd['state'][vi]['step']) + 1


def test_fused_optimizer_keeps_no_state_for_a_parameter_that_never_had_a_gradient():
  """torch.optim.AdamW creates state for a parameter when it first sees a gradient for it; a parameter that never had one (an
  unused head) has none, and a checkpoint written by FlatAdamW must look the same so that the reference's optimizer loads it unchanged."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  m = TD._model()
  opt = FlatAdamW(m.parameters(), lr=1e-3, amsgrad=True)
  w = normalized_loss_weights(m.config)
  plist = list(m.parameters())
  victim, other = m.target_speed_network[2].bias, m.target_speed_network[0].weight
  vi, oi = (next(i for i, q in enumerate(plist) if q is t) for t in (victim, other))
  for b in TD._batches(2):
    pred = m(rgb=b['rgb'], lidar_bev=b['lidar_bev'], target_point=b['target_point'], ego_vel=b['ego_vel'], command=b['command'])
    lab = {k: v for k, v in b.items() if k.endswith('_label')}
    ls = m.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                        pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
    sum(w[k] * v for k, v in ls.items()).backward()
    victim.grad = None
    opt.step()
    opt.zero_grad(set_to_none=True)
  sd = opt.state_dict()
  assert vi not in sd['state'] and oi in sd['state'] and float(sd['state'][oi]['step']) == 2.0
  ref = torch.optim.AdamW([p for p in plist], lr=1e-3, amsgrad=True)
  ref.load_state_dict(sd)  # the reference's optimizer takes the checkpoint as it is
  assert len(ref.state_dict()['state']) == len(sd['state'])


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def test_zero_redundancy_optimizer_and_ddp_with_learnable_weights_one_rank_rccl():
  """The reference's default optimizer wrapper (config.py:185 zero_redundancy_optimizer=1, train.py:527-529) around AdamW on the arena-backed
  parameters, and DistributedDataParallel with learn_multi_task_weights: the worker compares both with the plain loop."""
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'boundary_worker.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                     timeout=900, check=False)
  text = p.stdout.decode()
  assert p.returncode == 0, text[-4000:]
  r = json.loads([l for l in text.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):])
  TM._report('boundary_worker', r)
  assert max(r['zero_loss_rel']) < 2e-3 and r['zero_param_rel'] < 0.1, r
  assert r['ddp_learn_managed'] == 11 and r['ddp_learn_weight_moves'] == 10, r   # the anchor + the ten loss weights are DDP's
  assert max(r['ddp_learn_loss_rel']) < 2e-3, r
