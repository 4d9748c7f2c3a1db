#!/usr/bin/env python
"""Does a stream that waits on an EXTERNAL event recorded inside a captured hipGraph wait for THIS replay's record node?

The N > 1 training step keeps ONE captured graph and lets the RCCL stream start the all-reduce of a gradient bucket as soon as the bucket
is complete, i.e. in the middle of the replay: the capture records `torch.cuda.Event(external=True)` behind the kernels that finish a
bucket, and after `graph.replay()` a side stream calls `wait_event` on it.  That is only correct if the wait issued AFTER the launch
orders the side stream behind the record node of the replay in flight (and not behind a record of an earlier replay, or nothing).

Test: the graph runs  spin(~2 ms) -> flag = epoch -> [record external event] -> spin(~2 ms);  the side stream waits on the event and
copies the flag.  Correct: the copy sees this replay's epoch, and it completes BEFORE the graph's tail does (so the wait really is on
the node, not on the whole graph).  Prints one line per replay and a verdict."""
import sys
import time

import torch


def main():
  dev = torch.device('cuda')
  flag = torch.zeros(1, device=dev, dtype=torch.int64)
  epoch = torch.zeros(1, device=dev, dtype=torch.int64)
  seen = torch.zeros(1, device=dev, dtype=torch.int64)
  side = torch.cuda.Stream(dev)
  ev = torch.cuda.Event(external=True)
  spin = int(2e-3 * 2.0e9)  # ~2 ms at ~2 GHz
  torch.cuda._sleep(1000)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream(dev)
  with torch.cuda.graph(g, stream=cap):
    torch.cuda._sleep(spin)
    flag.copy_(epoch)
    ev.record(torch.cuda.current_stream())
    torch.cuda._sleep(spin)
  ok = True
  for it in range(1, 9):
    epoch.fill_(it)
    seen.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    side.wait_event(ev)
    with torch.cuda.stream(side):
      seen.copy_(flag)
      done_side = torch.cuda.Event()
      done_side.record(side)
    done_side.synchronize()
    t_side = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    v = int(seen.item())
    good = (v == it) and (t_side < 0.8 * t_all)
    ok &= good
    print(f'replay {it}: side stream saw epoch {v} after {t_side * 1e3:.2f} ms, graph finished after {t_all * 1e3:.2f} ms -> {"ok" if good else "WRONG"}', flush=True)
  print('EXTERNAL_EVENT_IN_GRAPH', 'OK' if ok else 'BROKEN', flush=True)
  sys.exit(0 if ok else 1)


if __name__ == '__main__':
  main()
