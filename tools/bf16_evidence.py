#!/usr/bin/env python
"""How close is the benchmarked precision (bf16 storage, fp32 accumulation) to the fp32 step?  (1) one step at bs = 12 on identical weights /
batch: per-tensor gradient-norm and sampled-element deviations; (2) 50 optimizer steps, bf16 Trainer beside fp32 Trainer, same seeds,
dropout off: the weighted training loss of both.  Numbers behind the bars of tests/test_model.py (printed as JSON)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import tfpp_port as P  # noqa: E402


def main():
  import test_model as T
  from carla_garage_amd.trainer import Trainer
  out = {}
  # ---- (1) one step, bs = 12
  m32 = T._model('fp32').train()
  _, v32, e32 = T._engine_train_step(m32, 12)
  ref = {n: e32.grads[n].detach().clone() for n in e32.grads}
  del m32, e32
  torch.cuda.empty_cache()
  m16 = T._model('bf16').train()
  names, v16, e16 = T._engine_train_step(m16, 12)
  big = max(float(g.double().norm()) for g in ref.values())
  rows = []
  for n, r in ref.items():
    rn = float(r.double().norm())
    if rn < 1e-3 * big:
      continue
    g = e16.grads[n].detach()
    idx = torch.from_numpy(T.U.sample_idx(r.numel())).to(r.device)
    rs, gs = r.flatten()[idx].double(), g.flatten()[idx].double()
    rms = rn / np.sqrt(r.numel())
    rows.append((n, abs(float(g.double().norm()) - rn) / rn, float(((gs - rs).abs() / (rms + rs.abs())).max())))
  canc = ('.se.fc', '.attn.query.', '.attn.key.')
  nerr = np.array([r[1] for r in rows])
  flat32 = torch.cat([ref[n].flatten().double() for n in ref])
  flat16 = torch.cat([e16.grads[n].detach().flatten().double() for n in ref])
  bad_el = good_el = 0
  for n, r in ref.items():
    rn = float(r.double().norm())
    if rn < 1e-3 * big:
      continue
    idx = torch.from_numpy(T.U.sample_idx(r.numel())).to(r.device)
    rs, gs = r.flatten()[idx].double(), e16.grads[n].detach().flatten()[idx].double()
    rel = (gs - rs).abs() / (rn / np.sqrt(r.numel()) + rs.abs())
    bad_el += int((rel > 0.5).sum())
    good_el += int((rel <= 0.5).sum())
  out['one_step_robust'] = {'arena_cosine': float((flat32 * flat16).sum() / (flat32.norm() * flat16.norm())), 'arena_rel_l2': float((flat16 - flat32).norm() / flat32.norm()),
                            'norm_err_median': float(np.median(nerr)), 'norm_err_p90': float(np.percentile(nerr, 90)), 'norm_err_p99': float(np.percentile(nerr, 99)),
                            'norm_err_max': float(nerr.max()), 'sampled_elements': bad_el + good_el, 'sampled_elements_beyond_0.5': bad_el}
  out['one_step'] = {'tensors': len(rows), 'loss_rel': {n: float(abs(a - b) / abs(b)) for n, a, b in zip(names, v16, v32)},
                     'norm_worst_plain': max(r[1] for r in rows if not any(c in r[0] for c in canc)),
                     'norm_worst_cancelling': max(r[1] for r in rows if any(c in r[0] for c in canc)),
                     'elem_worst_plain': max(r[2] for r in rows if not any(c in r[0] for c in canc)),
                     'elem_worst_cancelling': max(r[2] for r in rows if any(c in r[0] for c in canc)),
                     'norm_top': sorted(((round(r[1], 4), r[0]) for r in rows), reverse=True)[:6],
                     'elem_top': sorted(((round(r[2], 4), r[0]) for r in rows), reverse=True)[:6]}
  del m16, e16
  torch.cuda.empty_cache()
  # ---- (2) 50 steps, 4 different batches cycled
  batches = []
  for i in range(4):
    b = {k: v.cuda() for k, v in P.make_labels(4).items()}
    for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(4)):
      b[k] = v.cuda()
    b['rgb'] = (b['rgb'] + 5.0 * i).clamp(0, 255)
    batches.append(b)
  curves = {}
  for dt in ('fp32', 'bf16'):
    m = T._model(dt).train()
    T._zero_dropout(m)
    tr = Trainer(m, lr=1e-4)
    curves[dt] = [tr.total_loss(tr.train_step(batches[s % 4])) for s in range(50)]
    del tr, m
    torch.cuda.empty_cache()
  a, b = np.array(curves['fp32']), np.array(curves['bf16'])
  out['curve'] = {'fp32_first_last': [float(a[0]), float(a[-1])], 'bf16_first_last': [float(b[0]), float(b[-1])],
                  'max_rel_dev': float(np.max(np.abs(a - b) / np.abs(a))), 'mean_rel_dev': float(np.mean(np.abs(a - b) / np.abs(a))),
                  'fp32': [round(float(x), 4) for x in a[::5]], 'bf16': [round(float(x), 4) for x in b[::5]]}
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  main()
