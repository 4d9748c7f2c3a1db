#!/usr/bin/env python
"""Which tape event is the FIRST whose tensor differs between replays of one captured bs = 12 training step?

  TFPP_DEBUG_NODE_HASH=1 TFPP_SIDE_BATCH=128 python tools/replay_bisect.py [replays]

With TFPP_DEBUG_NODE_HASH=1 the engine hashes, inside the captured step, every tensor a forward primitive records and every gradient a
backward closure consumes / produces (ops.node_hash: an order-independent 64-bit integer sum per event, written by a kernel that is part
of the graph).  The table is read back after each replay and compared with the first replay's; events are listed in issue order, so the
first differing row names the kernel whose result depends on what ran beside it.  TFPP_DEBUG_NODE_HASH_RANGE=a:b hashes only events
a..b-1 (fewer extra launches = less perturbation of the schedule being diagnosed)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('TFPP_DEBUG_NODE_HASH', '1')
import torch  # noqa: E402

from tools.stress_step import make  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402


def main():
  replays = int(sys.argv[1]) if len(sys.argv) > 1 else 30
  tr, batch = make(int(os.environ.get('DIAG_BS', '12')), 'bf16', True)
  gs = GraphedTrainStep(tr, batch, warmup=1)
  st = ops.NODE_HASH
  gs()
  torch.cuda.synchronize()
  h0 = st['buf'].cpu().clone()
  g0 = tr.eng.flat_grad.clone()
  labels = list(st['labels'])
  ever = torch.zeros_like(h0, dtype=torch.bool)
  firsts, deviating, grad_dev = [], 0, 0
  for _ in range(replays):
    gs()
    torch.cuda.synchronize()
    d = st['buf'].cpu() != h0
    if bool((tr.eng.flat_grad != g0).any()):
      grad_dev += 1
    if bool(d.any()):
      deviating += 1
      firsts.append(int(d.nonzero()[0]))
      ever |= d
  n = len(labels)
  print(f'events {n} (hashed {st["lo"]}..{min(st["hi"], n)}), side batch {tr.eng.side.batch}, replays {replays}: hash tables deviating {deviating}, '
        f'gradient arenas deviating {grad_dev}')
  if deviating:
    print('first differing event per deviating replay:', sorted(set(firsts)), 'counts', {f: firsts.count(f) for f in sorted(set(firsts))})
    idx = [int(i) for i in ever.nonzero().flatten()[:int(os.environ.get('DIAG_TOP', '60'))]]
    lo = max(0, idx[0] - 12)
    print(f'--- context: events {lo}..{idx[0]} (equal in every replay)')
    for i in range(lo, idx[0]):
      print(f'      {i:5d} {labels[i] if i < n else "?"}')
    print('--- events around the first differing one')
    for i in range(idx[0], min(n, idx[0] + int(os.environ.get('DIAG_AFTER', '30')))):
      print(f'  {"DIFF" if bool(ever[i]) else "same"} {i:5d} {labels[i]}')
    if os.environ.get('DIAG_LIST', '0') == '1':
      print('--- differing events in issue order')
      for i in idx:
        print(f'  DIFF {i:5d} {labels[i] if i < n else "?"}')
    nb = sum(1 for i in range(n) if labels[i].startswith('bwd'))
    print(f'differing events total {int(ever.sum())}; forward events {n - nb}, backward events {nb}')


if __name__ == '__main__':
  main()
