#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM kernels on the model's dominant shapes (bs=12).  Used with rocprofv3 --pmc to
attribute time (profiles/).  Launches are replayed from a hipGraph, so the numbers are GPU time, not the Python launch rate."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402

SHAPES = [
    # name, B, H, W, Cin, Cout, k, stride, groups
    ('fusion_mlp0 3840x6048x1512', 3840, 1, 1, 1512, 6048, 1, 1, 1),
    ('fusion_mlp2 3840x1512x6048', 3840, 1, 1, 6048, 1512, 1, 1, 1),
    ('fusion_proj 3840x1512x1512', 3840, 1, 1, 1512, 1512, 1, 1, 1),
    ('fusion576_mlp0 3840x2304x576', 3840, 1, 1, 576, 2304, 1, 1, 1),
    ('s4_1x1 3072x1512x1512', 12, 8, 32, 1512, 1512, 1, 1, 1),
    ('s3_1x1 12288x576x576', 12, 16, 64, 576, 576, 1, 1, 1),
    ('lid_s3_1x1 3072x576x576', 12, 16, 16, 576, 576, 1, 1, 1),
    ('lid_s4_1x1 768x1512x1512', 12, 8, 8, 1512, 1512, 1, 1, 1),
    ('s2_1x1 49152x216x216', 12, 32, 128, 216, 216, 1, 1, 1),
    ('s1_1x1 196608x72x72', 12, 64, 256, 72, 72, 1, 1, 1),
    ('s1_in 786432x72x32', 12, 128, 512, 32, 72, 1, 1, 1),
    ('s1_conv3 196608x72x216?', 12, 64, 256, 216, 72, 1, 1, 1),
    ('s3_g3x3 12288 g24', 12, 16, 64, 576, 576, 3, 1, 24),
    ('dec_3x3 3.1Mx32x288', 12, 256, 1024, 32, 32, 3, 1, 1),
    ('head_3x3 49152x64x576', 12, 64, 64, 64, 64, 3, 1, 1),
]


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--iters', type=int, default=10)
  ap.add_argument('--only', default='')
  ap.add_argument('--wgrad', action='store_true')
  ap.add_argument('--dgrad', action='store_true')
  ap.add_argument('--eager', action='store_true', help='plain launches, no hipGraph (for rocprofv3 --pmc passes)')
  ap.add_argument('--shape', action='append', default=[], help='B,H,W,Cin,Cout,k,stride,G (repeatable; replaces the built-in list)')
  args = ap.parse_args()
  dev = 'cuda'
  dt = torch.bfloat16
  shapes = SHAPES
  if args.shape:
    shapes = [(f'custom {t}',) + tuple(int(v) for v in t.split(',')) for t in args.shape]
  for name, B, H, W, Cin, Cout, k, st, G in shapes:
    if args.only and args.only not in name:
      continue
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    x = (torch.rand(B, H, W, Cin, device=dev) - 0.5).to(dt)
    w = (torch.rand(Cout, Cin // G, k, k, device=dev) - 0.5) * 0.1
    wp = ops.pack_conv_weight(w, dt, G=G)
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=dt)
    dw = torch.zeros_like(w)
    flops = 2.0 * B * Ho * Wo * Cout * (Cin // G) * k * k

    def run():
      if args.wgrad:
        ops.conv_wgrad(y, x, dw, B=B, Hs=H, Ws=W, Cs=Cin, Hd=Ho, Wd=Wo, Cd=Cout, R=k, S=k, stride=st, pad=pad, G=G)
      else:
        ops.conv_gemm(x, wp, y, B=B, Hs=H, Ws=W, Cs=Cin, Hd=Ho, Wd=Wo, Cd=Cout, R=k, S=k, stride=st, pad=pad, G=G)

    run()
    torch.cuda.synchronize()
    if args.eager:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.iters):
        run()
      e1.record()
      torch.cuda.synchronize()
      print(f'eager {name:32s} {e0.elapsed_time(e1) / args.iters * 1e3:9.1f} us', flush=True)
      continue
    # replay `iters` launches from a hipGraph: measures GPU time, not the Python launch rate
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      for _ in range(args.iters):
        run()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    geo = dict(B=B, Hs=H, Ws=W, Cs=Cin, Hd=Ho, Wd=Wo, Cd=Cout, R=k, S=k, stride=st, pad=pad, G=G)
    plan = ops.conv_wgrad_plan(y, x, dw, **geo) if args.wgrad else ops.conv_gemm(x, wp, y, plan_only=True, **geo)
    print(f'{"wgrad" if args.wgrad else "conv "} {name:32s} {ms * 1e3:9.1f} us  {flops / ms / 1e9:8.1f} TFLOP/s  plan {plan}', flush=True)


if __name__ == '__main__':
  main()
