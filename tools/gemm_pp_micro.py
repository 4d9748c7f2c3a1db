#!/usr/bin/env python
"""Correctness sweep + micro-benchmark of the ping-pong GEMM (csrc/gemm_pp.hip) against the LDS-DMA ring kernels it replaces.

  python tools/gemm_pp_micro.py --check          every tile configuration x {K slices} on edge-case shapes vs an fp32 torch product
  python tools/gemm_pp_micro.py --bench          TFLOP/s of every configuration on the fusion-transformer / stage-4 shapes (hipGraph replays)

tfpp_gemm_pp_config(cfg): 0 / -1 off (ring kernels; the default), -2 automatic plan, 1 + i + 100 * s = configuration i with s K slices."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd import _lib  # noqa: E402

CFG_NAMES = ['256x256', '256x256/32', '256x192', '256x128', '128x256', '128x192', '128x128', '256x192 sr64']

BENCH_SHAPES = [  # name, M, N, K
    ('fusion_mlp0', 3840, 6048, 1512),
    ('fusion_mlp2', 3840, 1512, 6048),
    ('fusion_qkv', 3840, 4608, 1512),
    ('fusion_proj', 3840, 1512, 1512),
    ('s4_1x1', 3072, 1512, 1512),
    ('fusion576_mlp0', 3840, 2304, 576),
    ('fusion576_mlp2', 3840, 576, 2304),
    ('s3_1x1', 12288, 576, 576),
]


def set_cfg(v):
  rc = _lib.lib.raw('tfpp_gemm_pp_config')(int(v))
  assert rc == 0


def run_gemm(x, wp, y, M, N, K, **kw):
  return ops.conv_gemm(x, wp, y, B=M, Hs=1, Ws=1, Cs=K, Hd=1, Wd=1, Cd=N, **kw)


def check():
  dev, dt = 'cuda', torch.bfloat16
  torch.manual_seed(0)
  bad = 0
  shapes = [(256, 256, 64), (256, 256, 128), (256, 256, 192), (300, 200, 72), (512, 384, 1512), (1000, 1512, 520), (3840, 6048, 1512), (777, 1000, 4096),
            (128, 128, 64), (3072, 1512, 1512)]
  for (M, N, K) in shapes:
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.2)
    wp = ops.pack_conv_weight(w, dt, G=1)
    scale = torch.rand(N, device=dev) + 0.5
    shift = torch.rand(N, device=dev) - 0.5
    res = (torch.rand(M, N, device=dev) - 0.5).to(dt)
    ref0 = x.float() @ w.view(N, K).to(dt).float().t()
    for ci in range(len(CFG_NAMES)):
      for sp in (0, 2, 3):
        for epi in (0, 1):
          set_cfg(1 + ci + 100 * sp)
          y = torch.full((M, N), float('nan'), device=dev, dtype=dt)
          if epi:
            var, splits = run_gemm(x, wp, y, M, N, K, scale=scale, shift=shift, res=res, act=ops.ACT_RELU, alpha=0.5, plan_only=True)
            run_gemm(x, wp, y, M, N, K, scale=scale, shift=shift, res=res, act=ops.ACT_RELU, alpha=0.5)
            ref = torch.relu(0.5 * ref0 * scale + shift + res.float())
          else:
            var, splits = run_gemm(x, wp, y, M, N, K, plan_only=True)
            run_gemm(x, wp, y, M, N, K)
            ref = ref0
          torch.cuda.synchronize()
          err = (y.float() - ref).abs().max().item()
          tol = 0.02 * max(1.0, ref.abs().max().item())
          ok = (err <= tol) and bool(torch.isfinite(y.float()).all())
          if not ok or var != 210 + ci:
            bad += 1
            print(f'FAIL M={M} N={N} K={K} cfg={CFG_NAMES[ci]} splits_req={sp} plan=({var},{splits}) epi={epi} max_err={err:.4g} tol={tol:.3g}', flush=True)
    print(f'checked M={M} N={N} K={K}', flush=True)
  set_cfg(0)
  print('CHECK', 'FAILED' if bad else 'OK', bad, flush=True)
  return bad


def repeat_check(n=20):
  """race screen: the same launch repeated must reproduce bit for bit (LDS hazards show up as rare differing tiles)."""
  dev, dt = 'cuda', torch.bfloat16
  torch.manual_seed(1)
  bad = 0
  for (M, N, K) in [(3840, 6048, 1512), (3840, 1512, 6048)]:
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.2)
    wp = ops.pack_conv_weight(w, dt, G=1)
    for ci in range(len(CFG_NAMES)):
      set_cfg(1 + ci)
      y0 = torch.empty(M, N, device=dev, dtype=dt)
      run_gemm(x, wp, y0, M, N, K)
      for _ in range(n):
        y = torch.empty(M, N, device=dev, dtype=dt)
        run_gemm(x, wp, y, M, N, K)
        if not torch.equal(y, y0):
          bad += 1
          print(f'REPEAT MISMATCH M={M} N={N} K={K} cfg={CFG_NAMES[ci]} ndiff={(y != y0).sum().item()}', flush=True)
          break
  set_cfg(0)
  print('REPEAT', 'FAILED' if bad else 'OK', flush=True)
  return bad


def time_graph(fn, iters):
  fn()
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    for _ in range(iters):
      fn()
  graph.replay()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
  return best


def label_of(cfg):
  if cfg == -1:
    return 'ring'
  if cfg == -2:
    return 'auto'
  if cfg == 0:
    return 'ring'
  dbg, c = cfg // 10000, cfg % 10000
  return f'{CFG_NAMES[(c % 100) - 1]} s{c // 100}' + (f' dbg{dbg}' if dbg else '')


def eager(iters, only, cfgs):
  """plain launches (rocprofv3 --pmc passes: one configuration per process so the counters of a kernel name are one variant's)."""
  dev, dt = 'cuda', torch.bfloat16
  torch.manual_seed(0)
  for name, M, N, K in BENCH_SHAPES:
    if only and only not in name:
      continue
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.2)
    wp = ops.pack_conv_weight(w, dt, G=1)
    y = torch.empty(M, N, device=dev, dtype=dt)
    for cfg in cfgs:
      set_cfg(cfg)
      for _ in range(iters):
        run_gemm(x, wp, y, M, N, K)
      torch.cuda.synchronize()
      print('eager', name, label_of(cfg), flush=True)
  set_cfg(0)


def bench(iters, only, cfgs, rounds=5):
  """TFLOP/s of every configuration, hipGraph replays of `iters` launches, `rounds` interleaved rounds (min and median: box clocks drift)."""
  dev, dt = 'cuda', torch.bfloat16
  torch.manual_seed(0)
  for name, M, N, K in BENCH_SHAPES:
    if only and only not in name:
      continue
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.2)
    wp = ops.pack_conv_weight(w, dt, G=1)
    y = torch.empty(M, N, device=dev, dtype=dt)
    flops = 2.0 * M * N * K
    graphs = []
    for cfg in cfgs:
      set_cfg(cfg)
      plan = run_gemm(x, wp, y, M, N, K, plan_only=True)
      run_gemm(x, wp, y, M, N, K)
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        for _ in range(iters):
          run_gemm(x, wp, y, M, N, K)
      g.replay()
      graphs.append((cfg, plan, g, []))
    torch.cuda.synchronize()
    for _ in range(rounds):
      for cfg, plan, g, ts in graphs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    for cfg, plan, g, ts in graphs:
      ts = sorted(ts)
      best, med = ts[0], ts[len(ts) // 2]
      print(f'{name:16s} {M}x{N}x{K:5d} {label_of(cfg):18s} min {best * 1e3:7.1f} us {flops / best / 1e9:7.1f} TFLOP/s   median {med * 1e3:7.1f} us {flops / med / 1e9:7.1f} TFLOP/s  plan {plan}', flush=True)
    print(flush=True)
  set_cfg(0)


def cold(iters, only, cfgs, rounds=3):
  """The same comparison with COLD caches: a 1 GB fill between the launches evicts the operands from the L2s and the memory-side cache
  (in the training step the weights were last touched ~10 ms and ~500 MB of traffic ago).  Time = graph of [fill, GEMM] x iters minus graph of
  [fill] x iters."""
  dev, dt = 'cuda', torch.bfloat16
  torch.manual_seed(0)
  junk = torch.empty(256 << 20, device=dev, dtype=torch.float32)
  for name, M, N, K in BENCH_SHAPES:
    if only and only not in name:
      continue
    x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
    w = ((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.2)
    wp = ops.pack_conv_weight(w, dt, G=1)
    y = torch.empty(M, N, device=dev, dtype=dt)
    flops = 2.0 * M * N * K

    def graph_of(fn):
      fn()
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        for _ in range(iters):
          ops.zero_(junk)
          fn()
      g.replay()
      torch.cuda.synchronize()
      return g

    def t_of(g):
      best = 1e9
      for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
      return best

    base = t_of(graph_of(lambda: None))
    for cfg in cfgs:
      set_cfg(cfg)
      plan = run_gemm(x, wp, y, M, N, K, plan_only=True)
      ms = t_of(graph_of(lambda: run_gemm(x, wp, y, M, N, K))) - base
      print(f'cold {name:16s} {M}x{N}x{K:5d} {label_of(cfg):18s} {ms * 1e3:7.1f} us {flops / ms / 1e9:7.1f} TFLOP/s  plan {plan}  (fill alone {base * 1e3:.1f} us)', flush=True)
    print(flush=True)
  set_cfg(0)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--check', action='store_true')
  ap.add_argument('--repeat', action='store_true')
  ap.add_argument('--bench', action='store_true')
  ap.add_argument('--eager', action='store_true')
  ap.add_argument('--cold', action='store_true')
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--only', default='')
  ap.add_argument('--cfgs', default='')
  args = ap.parse_args()
  rc = 0
  if args.check:
    rc |= check()
  if args.repeat:
    rc |= repeat_check()
  if args.eager:
    eager(args.iters, args.only, [int(v) for v in args.cfgs.split(',')])
  if args.cold:
    cold(args.iters, args.only, [int(v) for v in args.cfgs.split(',')] if args.cfgs else [-1, -2, 3, 6, 7])
  if args.bench:
    cfgs = [int(v) for v in args.cfgs.split(',')] if args.cfgs else [-1, -2] + [1 + i for i in range(len(CFG_NAMES))]
    bench(args.iters, args.only, cfgs)
  sys.exit(1 if rc else 0)


if __name__ == '__main__':
  main()
