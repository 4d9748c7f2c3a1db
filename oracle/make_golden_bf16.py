"""bf16 reference for the benchmarked precision: the UNMODIFIED reference under ``torch.autocast`` (bfloat16).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``team_code/train.py:885`` wraps forward + compute_loss in
``torch.cuda.amp.autocast(enabled=bool(config.use_amp))``; with a bfloat16 autocast that is what "the reference in bf16"
computes: convolutions / linears / matmuls on bf16-rounded operands with fp32 accumulation, everything else (BatchNorm
statistics, softmax, losses) in fp32.  This script runs the reference (``oracle/ref_harness.py``: /root/reference imported
unmodified, build container only) twice on the deterministic test weights and the bs = 12 batch of the other goldens --
once in fp32, once under ``torch.autocast('cpu', dtype=torch.bfloat16)`` -- and writes

  tests/golden/tfpp_bf16_autocast_bs12.npz
      losses_fp32 / losses_autocast            the 10 losses
      stats_autocast_vs_fp32                   arena cosine, relative L2, per-tensor norm-error median / p90 / p99 / max,
                                               fraction of sampled elements beyond 0.5 x (rms + |ref|)  (tests/test_model.py metrics)
      grad_names / autocast_grad_norms / autocast_grad_samples      the autocast gradients themselves (norms + 16 samples per tensor)
      fp32_grad_norms                                                the reference's fp32 norms on the same weights (= tfpp_train_bs12.npz)

The GPU test (tests/test_model.py::test_bf16_step_is_no_worse_than_the_autocast_reference) holds the HIP bf16 step -- against the HIP
fp32 step, same statistics -- to these numbers.  The same comparison on the weights a bench run ends with is made at run time by
bench.py with the travelling port (oracle/tfpp_port.py) under the same autocast; tests/test_oracle.py pins port-under-autocast against
this fixture.

  python -m oracle.make_golden_bf16          # default configuration, bs = 12
  python -m oracle.make_golden_bf16 swin     # BASELINE config 5 (Video-Swin LiDAR branch), bs = 4 -> tests/golden/tfpp_swin_bf16_autocast_bs4.npz
"""
import os
import sys

import numpy as np
import torch

from oracle import ref_harness
from oracle import tfpp_port as P
from oracle.grad_stats import STAT_KEYS, gradient_stats
from oracle.make_golden import GOLDEN, GRAD_SAMPLES, disable_dropout, reference_loss_kwargs, sample_idx

def reference_step(model, cfg, bs, autocast):
  model.train()
  model.zero_grad(set_to_none=True)
  disable_dropout(model)
  inp, lab = P.make_inputs(bs, cfg), P.make_labels(bs, cfg)
  with torch.autocast('cpu', dtype=torch.bfloat16, enabled=autocast):  # (train.py:885 does this around forward + compute_loss)
    out = model(*inp)
    losses = model.compute_loss(**reference_loss_kwargs(out, lab))
  w = P.loss_weights(cfg)
  total = sum(w[k] * v.float() for k, v in losses.items())
  total.backward()
  grads = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
  return {k: float(v) for k, v in losses.items()}, grads


def main():
  if not ref_harness.available():
    sys.exit('needs /root/reference (build container)')
  torch.set_num_threads(os.cpu_count())
  swin = len(sys.argv) > 1 and sys.argv[1] == 'swin'
  if swin:
    # BASELINE config 5 at the setting bench.py times (Video-Swin LiDAR branch, 6 LiDAR frames, bs = 4): tests/golden/tfpp_swin_bf16_autocast_bs4.npz,
    # same contents; tests/test_model.py::test_video_swin_bf16_step_at_the_benchmarked_batch_is_no_worse_than_the_autocast_reference
    import dataclasses
    from oracle.make_golden import SWIN_OVERRIDES
    bs, fname = 4, 'tfpp_swin_bf16_autocast_bs4.npz'
    model, _ = ref_harness.build_reference_model(**SWIN_OVERRIDES)
    cfg = dataclasses.replace(P.PortConfig(), lidar_seq_len=6)
    sd = P.generic_state_dict(model.state_dict(), base=P.make_state_dict(P.PortConfig()))
  else:
    bs, fname = 12, 'tfpp_bf16_autocast_bs12.npz'
    model, _ = ref_harness.build_reference_model()
    cfg = P.PortConfig()
    sd = P.make_state_dict(cfg)
  model.load_state_dict(sd, strict=True)
  l32, g32 = reference_step(model, cfg, bs, False)
  model.load_state_dict(sd, strict=True)  # (the first step updated the BN running statistics)
  l16, g16 = reference_step(model, cfg, bs, True)
  st = gradient_stats(g32, g16)
  names = list(g16.keys())
  samples = np.zeros((len(names), GRAD_SAMPLES), np.float32)
  for i, n in enumerate(names):
    idx = sample_idx(g16[n].numel())
    samples[i, :len(idx)] = g16[n].flatten()[idx].numpy()
  d = {'loss_names': np.array(list(l32.keys())), 'losses_fp32': np.array(list(l32.values())), 'losses_autocast': np.array([l16[k] for k in l32]),
       'stat_names': np.array(STAT_KEYS), 'stats_autocast_vs_fp32': np.array([st[k] for k in STAT_KEYS]), 'tensors_compared': np.array(st['tensors']),
       'grad_names': np.array(names), 'autocast_grad_norms': np.array([float(g16[n].double().norm()) for n in names]),
       'fp32_grad_norms': np.array([float(g32[n].double().norm()) for n in names]), 'autocast_grad_samples': samples,
       'torch_version': np.array(torch.__version__), 'batch': np.array(bs)}
  np.savez_compressed(os.path.join(GOLDEN, fname), **d)
  print('losses fp32    ', l32)
  print('losses autocast', l16)
  print('autocast vs fp32 (reference, CPU):', st)


if __name__ == '__main__':
  main()
