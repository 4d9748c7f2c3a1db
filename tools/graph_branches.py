#!/usr/bin/env python
"""Does a hipGraph captured from two streams (fork/join with events) run its branches concurrently?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402


def main():
  dev, dt = 'cuda', torch.bfloat16
  M, K, N = 3072, 576, 576
  mk = lambda: ((torch.rand(M, 1, 1, K, device=dev) - 0.5).to(dt), ops.pack_conv_weight((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.1, dt),
                torch.empty(M, 1, 1, N, device=dev, dtype=dt))
  xa, wa, ya = mk()
  xb, wb, yb = mk()
  side = torch.cuda.Stream()
  n = 100

  def chain(x, w, y):
    for _ in range(n):
      ops.conv_gemm(x, w, y, B=M, Hs=1, Ws=1, Cs=K, Hd=1, Wd=1, Cd=N)

  chain(xa, wa, ya)
  with torch.cuda.stream(side):
    chain(xb, wb, yb)
  torch.cuda.synchronize()

  def capture(two_streams):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      main = torch.cuda.current_stream()
      if two_streams:
        side.wait_stream(main)
        with torch.cuda.stream(side):
          chain(xb, wb, yb)
        chain(xa, wa, ya)
        main.wait_stream(side)
      else:
        chain(xa, wa, ya)
        chain(xb, wb, yb)
    return g

  for two in (False, True):
    g = capture(two)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f'{"two streams" if two else "one stream "}: {e0.elapsed_time(e1) * 1e3 / (2 * n):7.2f} us per GEMM ({2 * n} launches)', flush=True)


if __name__ == '__main__':
  main()
