"""Debugging aid: which (stream, buffer) pairs do the split-K / weight-gradient-slice workspace and the other per-stream scratch tables hand out
during the eager warm-up and during the capture of the training step?"""
import collections
import sys

import torch

sys.path.insert(0, '.')
from tools.stress_step import make  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402

seen = collections.OrderedDict()
orig = ops.splitk_workspace


def logged(device):
  buf = orig(device)
  k = (torch.cuda.current_stream().cuda_stream, buf.data_ptr(), buf.numel(), bool(torch.cuda.is_current_stream_capturing()))
  seen[k] = seen.get(k, 0) + 1
  return buf


ops.splitk_workspace = logged
tr, batch = make(12, 'bf16', True)
tr.train_step(batch)
gs = GraphedTrainStep(tr, batch, warmup=1)
gs()
torch.cuda.synchronize()
for (st, ptr, n, cap), cnt in seen.items():
  print(f'stream {st:#x} buffer {ptr:#x} floats {n} capturing {cap} calls {cnt}')
for name, table in (('splitk', ops._SPLITK_WS), ('bn', ops._BN_SCRATCH), ('stats_rows', ops._STATS_ROWS), ('reduce', ops._REDUCE_SCRATCH)):
  print(name, [(k[0], hex(k[1]), hex(v.data_ptr()), v.numel()) for k, v in table.items()])
