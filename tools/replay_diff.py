"""Which parameter gradients differ between replays of the captured bs = 12 step (lr = 0, no dropout)?  Diagnostic for the stress test:
prints the parameters whose gradient slice is not bit-identical across replays, with the worst relative difference."""
import sys

import torch

sys.path.insert(0, '.')
from tools.stress_step import make  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd.engine import arena_order  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402


def main(replays=40):
  import os
  tr, batch = make(12, 'bf16', True)
  if os.environ.get('DIAG_EAGER', '0') == '1':
    tr.train_step(batch)
    gs = lambda: tr.train_step(batch)
  else:
    gs = GraphedTrainStep(tr, batch, warmup=1)
  gs()
  g0 = tr.eng.flat_grad.clone()
  worst = torch.zeros_like(g0)
  for _ in range(replays):
    gs()
    worst = torch.maximum(worst, (tr.eng.flat_grad - g0).abs())
  torch.cuda.synchronize()
  off, rows = 0, []
  for name, p in arena_order(tr.model)[0]:
    n = p.numel()
    d = float(worst[off:off + n].max())
    if d > 0:
      rows.append((d / (float(g0[off:off + n].abs().max()) + 1e-30), name, n, int((worst[off:off + n] > 0).sum()), d, float(g0[off:off + n].abs().max())))
    off += ops.pad_to(n, 4)
  import collections
  groups = collections.Counter('.'.join(r[1].split('.')[:3]) for r in rows)
  print('by module:', dict(groups))
  if os.environ.get('DIAG_SORT', 'rel') == 'rel':
    rows.sort(reverse=True)
  print(len(rows), 'parameters differ')
  for r in rows[:int(os.environ.get('DIAG_TOP', '40'))]:
    print(f'{r[0]:.3e} {r[1]:70s} numel {r[2]:9d} differing {r[3]:9d} abs {r[4]:.3e} max|g| {r[5]:.3e}')


if __name__ == '__main__':
  main()
