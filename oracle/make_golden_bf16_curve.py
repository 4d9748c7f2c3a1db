"""Training curves of the bf16 reference: the oracle port (pinned to the unmodified reference, tests/test_oracle.py) trained for 200 optimizer steps
in fp32 and under ``torch.autocast('cpu', bfloat16)`` -- AdamW(amsgrad), lr 1e-4, four batches of 4 cycled, dropout off: exactly the schedule of
tests/test_model.py::test_bf16_trains_like_fp32 -- written to tests/golden/tfpp_bf16_autocast_curve.npz.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The fixture answers "how far does a bf16 run of THIS network drift from its fp32 run over 200
steps when the reference's own arithmetic does it?": the HIP bf16 run is held to that drift, not to an absolute number.

  python -m oracle.make_golden_bf16_curve [steps]          (~20 min on 8 cores)
"""
import dataclasses
import os
import sys

import numpy as np
import torch

from oracle import tfpp_port as P
from oracle.make_golden import GOLDEN


def batches(cfg):
  out = []
  for i in range(4):
    inp = list(P.make_inputs(4, cfg))
    inp[0] = (inp[0] + 5.0 * i).clamp(0, 255)
    out.append((inp, P.make_labels(4, cfg)))
  return out


def curve(steps, autocast):
  cfg = dataclasses.replace(P.PortConfig(), embd_pdrop=0.0, resid_pdrop=0.0, attn_pdrop=0.0, decoder_dropout=0.0)
  frozen = lambda k: ('valid_bev' in k or 'running' in k or 'num_batches' in k or k.startswith('loss_'))
  sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not frozen(k) else v.clone()) for k, v in P.make_state_dict(cfg).items()}
  opt = torch.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=1e-4, amsgrad=True)
  data = batches(cfg)
  out = []
  for s in range(steps):
    inp, lab = data[s % 4]
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):
      o = P.forward(sd, cfg, *inp, training=True)
      total, _ = P.total_loss(sd, cfg, o, lab)
    opt.zero_grad(set_to_none=True)
    total.float().backward()
    opt.step()
    out.append(float(total))
    if s % 20 == 0:
      print(('autocast' if autocast else 'fp32'), s, out[-1], flush=True)
  return np.array(out)


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  torch.set_num_threads(os.cpu_count())
  a = curve(steps, False)
  b = curve(steps, True)
  np.savez_compressed(os.path.join(GOLDEN, 'tfpp_bf16_autocast_curve.npz'), fp32=a, autocast=b, steps=np.array(steps), torch_version=np.array(torch.__version__))
  d = np.abs(a - b) / np.abs(a)
  print('autocast vs fp32 over', steps, 'steps: max', d.max(), 'mean', d.mean(), 'last8', a[-8:].mean(), b[-8:].mean())


if __name__ == '__main__':
  main()
