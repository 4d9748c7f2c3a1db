#!/usr/bin/env python
"""bf16 vs fp32 training curves over N optimizer steps (AdamW-amsgrad, lr 1e-4, four batches of 4 cycled, dropout off), the numbers behind the bars of
tests/test_model.py::test_bf16_trains_like_fp32: raw and smoothed (window 8) relative deviation at several horizons.  Run it under two builds / switches
that only differ in the ORDER of fp32 sums (e.g. TFPP_HW_TICKET=0 / 1) to see the noise floor of these statistics.  usage: python tools/bf16_curve.py [steps]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import tfpp_port as P  # noqa: E402  (deterministic weights / inputs)


def main(steps=200):
  import test_model as T
  from carla_garage_amd.trainer import Trainer
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
    curves[dt] = np.array([tr.total_loss(tr.train_step(batches[s % 4])) for s in range(steps)])
    del tr, m
    torch.cuda.empty_cache()
  a, b = curves['fp32'], curves['bf16']
  smooth = lambda x: np.convolve(x, np.ones(8) / 8, mode='valid')
  out = {'steps': steps}
  for h in (50, 100, 200):
    if h > steps:
      continue
    d = np.abs(a[:h] - b[:h]) / np.abs(a[:h])
    ds = np.abs(smooth(a[:h]) - smooth(b[:h])) / np.abs(smooth(a[:h]))
    out[f'h{h}'] = {'raw_max': round(float(d.max()), 4), 'raw_mean': round(float(d.mean()), 4), 'smooth8_max': round(float(ds.max()), 4), 'smooth8_mean': round(float(ds.mean()), 4),
                    'last8_fp32': round(float(a[h - 8:h].mean()), 4), 'last8_bf16': round(float(b[h - 8:h].mean()), 4)}
  print(json.dumps(out))


if __name__ == '__main__':
  main(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
