#!/usr/bin/env python
"""Per-node cost of back-to-back dependent kernels replayed from a hipGraph (tiny kernels: the dispatch floor)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402


def timed_graph(fn, n):
  fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(n):
      fn()
  g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  g.replay()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3


def main():
  dev = 'cuda'
  for numel in (256, 65536, 1 << 20, 1 << 24):
    x = torch.zeros(numel, device=dev, dtype=torch.bfloat16)
    y = torch.zeros(numel, device=dev, dtype=torch.bfloat16)
    print(f'axpy  n={numel:9d}: {timed_graph(lambda: ops.axpy(x, y, 1.0), 200):7.2f} us/node', flush=True)
    print(f'zero  n={numel:9d}: {timed_graph(lambda: ops.zero_(y), 200):7.2f} us/node', flush=True)
  a = torch.zeros(1024, device=dev)
  print(f'torch add_ (1024 f32): {timed_graph(lambda: a.add_(1.0), 200):7.2f} us/node', flush=True)


if __name__ == '__main__':
  main()
