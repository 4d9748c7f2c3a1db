#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound BatchNorm / squeeze-excite / elementwise kernels on the activation shapes of the
RegNetY-3.2GF branches at bs=12 (image 256x1024, LiDAR BEV 256x256).  Prints achieved GB/s on ALGORITHMIC bytes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402

SHAPES = [  # name, B, H, W, C
    ('img_stem', 12, 128, 512, 32),
    ('img_s1', 12, 64, 256, 72),
    ('img_s2', 12, 32, 128, 216),
    ('img_s3', 12, 16, 64, 576),
    ('img_s4', 12, 8, 32, 1512),
    ('lid_s1', 12, 64, 64, 72),
    ('lid_s2', 12, 32, 32, 216),
    ('lid_s3', 12, 16, 16, 576),
    ('lid_s4', 12, 8, 8, 1512),
]


def timeit(fn, iters):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--only', default='')
  args = ap.parse_args()
  dev, dt = 'cuda', torch.bfloat16
  for name, B, H, W, C in SHAPES:
    if args.only and args.only not in name:
      continue
    x = (torch.rand(B, H, W, C, device=dev) - 0.5).to(dt)
    dy = (torch.rand(B, H, W, C, device=dev) - 0.5).to(dt)
    y = torch.empty_like(x)
    nbytes = x.numel() * 2
    scale = torch.rand(C, device=dev) + 0.5
    shift = torch.rand(C, device=dev) - 0.5
    gamma = torch.rand(C, device=dev) + 0.5
    mean = torch.zeros(C, device=dev)
    invstd = torch.ones(C, device=dev)
    ws = torch.zeros(4 * C, device=dev, dtype=torch.float64)
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    gate = torch.rand(B, C, device=dev)
    dpool = torch.rand(B, C, device=dev)
    colout = torch.zeros(C, device=dev)
    rows = B * H * W
    tests = [
        ('affine_act(relu)', lambda: ops.affine_act(x, y, scale=scale, shift=shift, act=ops.ACT_RELU), 2),
        ('affine_act(res,relu)', lambda: ops.affine_act(x, y, scale=scale, shift=shift, res=dy, act=ops.ACT_RELU), 3),
        ('affine_act(gate)', lambda: ops.affine_act(x, y, gate=gate, rows_per_batch=H * W), 2),
        ('bn_stats', lambda: ops.bn_stats(x, ws), 1),
        ('bn_bwd(reduce+apply)', lambda: ops.bn_bwd(dy, y, x, gamma, mean, invstd, ws, dgamma, dbeta, True), 6),
        ('mean_hw', lambda: ops.mean_hw(x), 1),
        ('se_dgate', lambda: ops.se_dgate(dy, x), 2),
        ('se_bwd_apply', lambda: ops.se_bwd_apply(dy, gate, dpool), 2),
        ('colsum', lambda: ops.colsum(x, colout, rows, C), 1),
        ('axpy', lambda: ops.axpy(x, y, 1.0), 3),
        ('zero', lambda: ops.zero_(y), 1),
    ]
    for tname, fn, passes in tests:
      us = timeit(fn, args.iters)
      print(f'{name:9s} [{rows:7d} x {C:4d}] {tname:22s} {us:8.1f} us  {passes * nbytes / us / 1e3:8.1f} GB/s', flush=True)


if __name__ == '__main__':
  main()
