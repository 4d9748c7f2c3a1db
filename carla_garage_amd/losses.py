"""Fused loss + gradient evaluation (team_code/model.py:394-445, team_code/center_net.py:77-123,
team_code/transfuser_utils.py:341-364) on the engine's internal NHWC tensors.

``fused_losses`` is the trainer path: one HIP kernel per loss writes the scalar and d(weight*loss)/d(pred) in a single
pass.  ``reference_form_losses`` backs ``LidarCenterNet.compute_loss`` for drop-in use under team_code/train.py: the
same kernels, wrapped so that each returned 0-d tensor is connected to the autograd graph of the forward outputs.
"""
import torch

from . import ops
from .engine import F32, EPS_F32

LOSS_ORDER = ('loss_wp', 'loss_selection', 'loss_target_speed', 'loss_checkpoint', 'loss_semantic', 'loss_bev_semantic', 'loss_depth',
              'loss_center_heatmap', 'loss_wh', 'loss_offset', 'loss_yaw_class', 'loss_yaw_res', 'loss_velocity', 'loss_brake')
BB_LOSSES = ('loss_center_heatmap', 'loss_wh', 'loss_offset', 'loss_yaw_class', 'loss_yaw_res', 'loss_velocity', 'loss_brake')


def temporal(cfg):
  """center_net.py:29-31,119-123: velocity / brake heads and losses exist when the input has more than one frame."""
  return not (cfg.lidar_seq_len == 1 and cfg.seq_len == 1)


def multi_wp(cfg):
  """config.multi_wp_output (config.py:484, model.py:151-163): two waypoint hypotheses, the better one per sample is trained, and a logit that
  learns which one that is (loss_selection, weight 1.0: train.py:440-441)."""
  return bool(cfg.use_wp_gru and getattr(cfg, 'multi_wp_output', False))


def active_losses(cfg):
  names = []
  if cfg.use_wp_gru:
    names.append('loss_wp')
    if multi_wp(cfg):
      names.append('loss_selection')
  if cfg.use_controller_input_prediction:
    names += ['loss_target_speed', 'loss_checkpoint']
  if cfg.use_semantic:
    names.append('loss_semantic')
  if cfg.use_bev_semantic:
    names.append('loss_bev_semantic')
  if cfg.use_depth:
    names.append('loss_depth')
  if cfg.detect_boxes:
    names += ['loss_center_heatmap', 'loss_wh', 'loss_offset', 'loss_yaw_class', 'loss_yaw_res']
    if temporal(cfg):
      names += ['loss_velocity', 'loss_brake']
  return names


def output_slots(cfg):
  """[(key, loss name)] of the caller-facing predictions in the order model._export emits them: one per active loss, except that loss_wp of
  the multi_wp_output variant has two (pred_wp under 'loss_wp', pred_wp_1 under 'loss_wp/1')."""
  slots = []
  for n in active_losses(cfg):
    slots.append((n, n))
    if n == 'loss_wp' and multi_wp(cfg):
      slots.append(('loss_wp/1', n))
  return slots


def normalized_loss_weights(cfg):
  """team_code/train.py:383-456: unused losses are zeroed, the remaining detailed_loss_weights (all 1.0 by default,
  team_code/config.py:223-239) are divided by their sum."""
  base = getattr(cfg, 'detailed_loss_weights', None)
  names = active_losses(cfg)
  w = {n: float(base[n]) if base is not None and n in base else 1.0 for n in names}
  s = sum(w.values())
  return {n: v / s for n, v in w.items()}


def _one_loss(model, name, t, labels, slot, weight, want_grad, af_sum):
  """Launch the kernel for one loss on internal tensors ``t``; returns (pred_tensor, dpred or None)."""
  cfg = model.config
  dev = slot.device
  smooth = float(cfg.label_smoothing_alpha) if getattr(cfg, 'use_label_smoothing', False) else 0.0  # model.py:252-265: the three class-weighted CE losses

  def grad_like(x):
    return torch.empty_like(x) if want_grad else None

  if name == 'loss_target_speed':
    p = t['pred_target_speed']  # [B, 8] fp32 (4 real)
    d = grad_like(p)
    ws = torch.empty(2, device=dev, dtype=F32)
    focal = bool(getattr(cfg, 'use_focal_loss', False))  # model.py:255-256: FocalLoss(alpha=speed_weights, gamma) instead of the (smoothed) cross entropy
    ops.ce_loss(p, labels['target_speed_label'], slot, ws, rows=p.shape[0], C=len(cfg.target_speeds), ld=p.shape[1], HW=p.shape[0],
                class_weight=model.loss_speed.nll_loss.weight if focal else model.loss_speed.weight, weight=weight, dpred=d,
                smoothing=0.0 if focal else smooth, focal_gamma=float(model.loss_speed.gamma) if focal else -1.0)
    return p, d
  if name == 'loss_wp' and multi_wp(cfg):
    p = t['pred_wp_pair']  # [B, 2, n, 2] fp32
    d = grad_like(p)
    t['_selection_labels'] = torch.empty(p.shape[0], device=dev, dtype=F32)  # the hypothesis that won, per sample: the target of loss_selection below
    ops.min_l1_pair_loss(p, labels['waypoint_label'].float().contiguous(), slot, t['_selection_labels'], weight=weight, dpair=d)
    return p, d
  if name == 'loss_selection':
    p = t['selected_path']  # [B, 8] fp32 (1 real)
    d = grad_like(p)
    ops.bce_logits_loss(p, t['_selection_labels'], slot, weight=weight, dlogit=d)
    return p, d
  if name in ('loss_checkpoint', 'loss_wp'):
    p = t['pred_checkpoint'] if name == 'loss_checkpoint' else t['pred_wp']
    lab = labels['checkpoint_label'] if name == 'loss_checkpoint' else labels['waypoint_label']
    d = grad_like(p)
    ops.reg_loss(p, lab.contiguous(), slot, B=p.shape[0], C=1, HW=p.shape[1] * p.shape[2], ld=1, kind=0, weight=weight, dpred=d)
    return p, d
  if name == 'loss_semantic':
    p = t['pred_semantic']  # [B,H,W,8]
    d = grad_like(p)
    B, H, W, ld = p.shape
    ws = torch.empty(2, device=dev, dtype=F32)
    ops.ce_loss(p, labels['semantic_label'], slot, ws, rows=B * H * W, C=cfg.num_semantic_classes, ld=ld, HW=H * W,
                class_weight=model.loss_semantic.weight, weight=weight, dpred=d, smoothing=smooth)
    return p, d
  if name == 'loss_bev_semantic':
    p = t['pred_bev_semantic']
    d = grad_like(p)
    B, H, W, ld = p.shape
    ws = torch.empty(2, device=dev, dtype=F32)
    ops.ce_loss(p, labels['bev_semantic_label'], slot, ws, rows=B * H * W, C=cfg.num_bev_semantic_classes, ld=ld, HW=H * W,
                class_weight=model.loss_bev_semantic.weight, vis_mask=model.valid_bev_pixels.detach().view(-1), weight=weight, dpred=d, smoothing=smooth)
    return p, d
  if name == 'loss_depth':
    p = t['pred_depth']  # sigmoid output [B,H,W,8]
    d = grad_like(p)
    B, H, W, ld = p.shape
    ops.reg_loss(p, labels['depth_label'].contiguous(), slot, B=B, C=1, HW=H * W, ld=ld, kind=0, weight=weight, dpred=d)
    return p, d
  # CenterNet head
  idx = BB_LOSSES.index(name)
  p = t['bb'][idx]
  d = grad_like(p)
  B, H, W, ld = p.shape
  pw = labels['pixel_weight_label'].contiguous()
  common = dict(B=B, HW=H * W, ld=ld, denom=af_sum, denom_eps=EPS_F32, weight=weight, dpred=d)
  if name == 'loss_center_heatmap':
    ops.reg_loss(p, labels['center_heatmap_label'].contiguous(), slot, C=cfg.num_bb_classes, kind=2, **common)
  elif name == 'loss_wh':
    ops.reg_loss(p, labels['wh_label'].contiguous(), slot, C=2, kind=0, elem_weight=pw, wC=2, denom_mul=2.0, **common)
  elif name == 'loss_offset':
    ops.reg_loss(p, labels['offset_label'].contiguous(), slot, C=2, kind=0, elem_weight=pw, wC=2, denom_mul=2.0, **common)
  elif name == 'loss_yaw_res':
    ops.reg_loss(p, labels['yaw_res_label'].contiguous(), slot, C=1, kind=1, elem_weight=pw, wC=2, w_bcast=True, **common)
  elif name == 'loss_velocity':  # L1 * pixel_weight[:, 0:1] / avg_factor (center_net.py:120)
    ops.reg_loss(p, labels['velocity_label'].contiguous(), slot, C=1, kind=0, elem_weight=pw, wC=2, w_bcast=True, **common)
  elif name == 'loss_brake':     # CE over 2 classes * pixel_weight[:, 0] / avg_factor (center_net.py:121)
    ws = torch.empty(2, device=dev, dtype=F32)
    ops.ce_loss(p, labels['brake_target_label'], slot, ws, rows=B * H * W, C=2, ld=ld, HW=H * W, pix_weight=pw,
                pw_bstride=2 * H * W, denom=af_sum, denom_eps=EPS_F32, weight=weight, dpred=d)
  else:
    ws = torch.empty(2, device=dev, dtype=F32)
    ops.ce_loss(p, labels['yaw_class_label'], slot, ws, rows=B * H * W, C=cfg.num_dir_bins, ld=ld, HW=H * W, pix_weight=pw,
                pw_bstride=2 * H * W, denom=af_sum, denom_eps=EPS_F32, weight=weight, dpred=d)
  return p, d


def fused_losses(model, t, labels, weights=None, want_grad=True):
  """Returns (names, loss_vector [n] fp32 of the UNWEIGHTED losses, seeds [(pred, dpred)] with the weights folded in)."""
  names = active_losses(model.config)
  dev = t['fused_features'].device
  vals = ops.zeros(len(names), F32, dev)
  af_sum = None
  if model.config.detect_boxes:
    af_sum = torch.empty(1, device=dev, dtype=F32)
    ops.sum_f32(labels['avg_factor_label'].float().contiguous(), af_sum)
  seeds = []
  for i, n in enumerate(names):
    w = 1.0 if weights is None else weights[n]
    p, d = _one_loss(model, n, t, labels, vals[i:i + 1], w, want_grad, af_sum)
    if want_grad:
      seeds.append((p, d))
  return names, vals, seeds


class _LossNode(torch.autograd.Function):
  """0-d loss connected to the caller-facing prediction tensor; backward returns dLoss/dpred in the caller layout.  (General path: the
  predictions handed to compute_loss are not the ones the last training forward returned; dropin.py has the fast one.)"""

  @staticmethod
  def forward(ctx, pred_caller, value, to_caller_grad):
    ctx.to_caller_grad = to_caller_grad
    return value.detach()

  @staticmethod
  def backward(ctx, g):
    return ctx.to_caller_grad(g), None, None


class _PairLossNode(torch.autograd.Function):
  """_LossNode for loss_wp of the multi_wp_output variant: one scalar, two caller-facing predictions (pred_wp, pred_wp_1)."""

  @staticmethod
  def forward(ctx, caller0, caller1, value, to_caller_grads):
    ctx.to_caller_grads = to_caller_grads
    return value.detach()

  @staticmethod
  def backward(ctx, g):
    g0, g1 = ctx.to_caller_grads(g)
    return g0, g1, None, None


def _scale_by_device_scalar(x, g):
  """x * g for a 0-d device tensor g without a host sync (broadcast g into a per-channel scale vector)."""
  ld = x.shape[-1] if x.dim() > 1 and x.shape[-1] % 4 == 0 else 4
  v = torch.empty(ld, device=x.device, dtype=F32)
  ops.copy_rows(g.float().reshape(1), v, ld, 1, 0, 0, 1, 0)
  return ops.affine_act(x.reshape(-1, ld), scale=v).view(x.shape)


def _internal_from_callers(model, args):
  """The engine's internal prediction tensors (NHWC, channel-padded, compute dtype) rebuilt from caller-facing ones (fp32 NCHW): lets
  compute_loss evaluate predictions that are NOT the outputs of the last forward (kept from an earlier step, cloned, post-processed),
  as the reference's compute_loss can (team_code/model.py:394-445)."""
  cfg, dt_ = model.config, model.compute_dtype
  t = dict(pred_wp=None, pred_target_speed=None, pred_checkpoint=None, pred_semantic=None, pred_bev_semantic=None, pred_depth=None, bb=None)

  def dense(x, pad):
    x = x.detach().float().contiguous()
    if x.dim() == 3:
      x = x.unsqueeze(1)
    return ops.nchw_to_nhwc_pad(x, dt_, ops.pad_to(x.shape[1], pad))

  if args.get('pred_wp') is not None and multi_wp(cfg):
    w0, w1 = (args[k].detach().float().contiguous() for k in ('pred_wp', 'pred_wp_1'))
    B, n = w0.shape[0], w0.shape[1] * w0.shape[2]
    t['pred_wp_pair'] = torch.empty((B, 2) + tuple(w0.shape[1:]), device=w0.device, dtype=F32)
    ops.copy_rows(w0, t['pred_wp_pair'], B, n, n, 0, 2 * n, 0)
    ops.copy_rows(w1, t['pred_wp_pair'], B, n, n, 0, 2 * n, n)
    sp = args['selected_path'].detach().float().contiguous()
    t['selected_path'] = ops.zeros((B, 8), F32, sp.device)
    ops.copy_rows(sp, t['selected_path'], B, 1, 1, 0, 8, 0)
  elif args.get('pred_wp') is not None:
    t['pred_wp'] = args['pred_wp'].detach().float().contiguous()
  if args.get('pred_target_speed') is not None:
    ts = args['pred_target_speed'].detach().float().contiguous()
    pad = ops.zeros((ts.shape[0], ops.pad_to(ts.shape[1], 8)), F32, ts.device)
    ops.copy_rows(ts, pad, ts.shape[0], ts.shape[1], ts.shape[1], 0, pad.shape[1], 0)
    t['pred_target_speed'] = pad
    t['pred_checkpoint'] = args['pred_checkpoint'].detach().float().contiguous()
  if args.get('pred_semantic') is not None:
    t['pred_semantic'] = dense(args['pred_semantic'], 8)
  if args.get('pred_bev_semantic') is not None:
    t['pred_bev_semantic'] = dense(args['pred_bev_semantic'], 8)
  if args.get('pred_depth') is not None:
    t['pred_depth'] = dense(args['pred_depth'], 8)
  if cfg.detect_boxes and args.get('pred_bounding_box') is not None:
    t['bb'] = [dense(b, 8) for b in args['pred_bounding_box'][:7 if temporal(cfg) else 5]]
  t['fused_features'] = next(v for v in (t['pred_checkpoint'], t['pred_wp'], t.get('pred_wp_pair'), t['pred_semantic']) if v is not None)  # (device carrier for fused_losses)
  return t


def reference_form_losses(model, args):
  """Drop-in ``compute_loss`` (team_code/model.py:394-445; called at team_code/train.py:784-820): the losses are evaluated by the fused HIP
  kernels.  Fast path: the predictions are the outputs of the last training forward (dropin.DropinStep.losses: internal tensors, token
  gradients, hipGraph replay).  General path: any other caller-facing predictions are converted to the internal layout first and the
  gradients take the caller layout on their way back through autograd."""
  cfg = model.config
  label_keys = ('waypoint_label', 'target_speed_label', 'checkpoint_label', 'semantic_label', 'bev_semantic_label', 'depth_label',
                'center_heatmap_label', 'wh_label', 'yaw_class_label', 'yaw_res_label', 'offset_label', 'velocity_label',
                'brake_target_label', 'pixel_weight_label', 'avg_factor_label')
  labels = {k: args[k] for k in label_keys if args.get(k) is not None}
  callers = {'loss_wp': args['pred_wp'], 'loss_target_speed': args['pred_target_speed'], 'loss_checkpoint': args['pred_checkpoint'],
             'loss_semantic': args['pred_semantic'], 'loss_bev_semantic': args['pred_bev_semantic'], 'loss_depth': args['pred_depth']}
  if multi_wp(cfg):
    callers['loss_wp/1'], callers['loss_selection'] = args['pred_wp_1'], args['selected_path']
  if cfg.detect_boxes:
    for i, n in enumerate(BB_LOSSES[:7 if temporal(cfg) else 5]):
      callers[n] = args['pred_bounding_box'][i]
  step = model.__dict__.get('_dropin_step')
  if step is not None and step.owns(callers):
    return step.losses(callers, labels)
  mine = model.__dict__.get('_last_output_ptrs', set())
  t = model.__dict__.get('_last_internal')
  if t is None or any(c is not None and c.data_ptr() not in mine for c in callers.values()):
    t = _internal_from_callers(model, args)  # not (all) outputs of the last forward
  want_grad = torch.is_grad_enabled() and any(c is not None and c.requires_grad for c in callers.values())
  names, vals, seeds = fused_losses(model, t, labels, None, want_grad)
  out = {}
  for i, n in enumerate(names):
    caller = callers[n]
    if want_grad and caller.requires_grad:
      pred, dpred = seeds[i]

      def to_caller(g, pred=pred, dpred=dpred, caller=caller, n=n):
        d = _scale_by_device_scalar(dpred, g)
        if n == 'loss_wp' and multi_wp(cfg):  # [B, 2, n, 2] -> the gradients of pred_wp and pred_wp_1
          B_, n_ = caller.shape[0], caller.shape[1] * caller.shape[2]
          halves = [torch.empty(caller.shape, device=d.device, dtype=F32) for _ in range(2)]
          for h, o in enumerate(halves):
            ops.copy_rows(d, o, B_, n_, 2 * n_, h * n_, n_, 0)
          return halves
        if n == 'loss_selection':
          o = torch.empty(caller.shape, device=d.device, dtype=F32)
          ops.copy_rows(d, o, caller.shape[0], 1, d.shape[1], 0, 1, 0)
          return o
        if n == 'loss_target_speed':
          o = torch.empty(caller.shape, device=d.device, dtype=F32)
          ops.copy_rows(d, o, caller.shape[0], caller.shape[1], d.shape[1], 0, caller.shape[1], 0)
          return o
        if d.dim() == 4:
          c_real = caller.shape[1] if caller.dim() == 4 else 1
          return ops.nhwc_to_nchw(d, c_real).view(caller.shape)
        return d.view(caller.shape)

      if n == 'loss_wp' and multi_wp(cfg):
        out[n] = _PairLossNode.apply(caller, callers['loss_wp/1'], vals[i], to_caller)
      else:
        out[n] = _LossNode.apply(caller, vals[i], to_caller)
    else:
      out[n] = vals[i].detach()
  return out
