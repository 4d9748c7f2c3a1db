"""Does a hipGraph keep the order of a LONG single-stream segment captured on a side stream while the capture stream runs other work?
Side branch: n dependent launches x = x * a + b on one buffer (any reordering or overlap changes the result) + a second buffer used as
scratch by every launch (write then read back), like the weight-gradient lane's shared workspaces.  Main branch: independent GEMMs.
Prints, per n, whether 20 replays reproduce the eager result bit for bit."""
import sys

import torch


def run(n, main_work=60):
  dev = torch.device('cuda:0')
  side = torch.cuda.Stream()
  x = torch.zeros(1 << 16, device=dev)
  scratch = torch.zeros(1 << 16, device=dev)
  a = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
  out = torch.empty_like(a)

  def body():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
      for i in range(n):
        torch.mul(x, 1.0001, out=scratch)       # scratch = f(x)
        torch.add(scratch, float(i % 7), out=x)  # x = g(scratch)
    for _ in range(main_work):
      torch.mm(a, a, out=out)
    cur.wait_stream(side)

  x.zero_()
  body()
  torch.cuda.synchronize()
  ref = x.clone()
  g = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream()
  cap.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(cap):
    x.zero_()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
      body()
  bad = 0
  for _ in range(20):
    x.zero_()
    g.replay()
    torch.cuda.synchronize()
    bad += int(not torch.equal(x, ref))
  return bad


if __name__ == '__main__':
  for n in [int(v) for v in sys.argv[1:]] or [32, 64, 128, 200, 400, 800]:
    print(f'side segment of {2 * n:5d} launches: {run(n)} of 20 replays differ from the eager result', flush=True)
