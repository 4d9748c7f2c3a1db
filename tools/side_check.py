"""Debugging aid: one eager bs = 12 training step with TFPP_DEBUG_SIDE_CHECK=1 (every tensor handed to the weight-gradient lane is
checksummed at hand-over and again at the join)."""
import os
import sys

os.environ['TFPP_DEBUG_SIDE_CHECK'] = '1'
sys.path.insert(0, '.')
from tools.stress_step import make  # noqa: E402

tr, batch = make(12, 'bf16', True)
for _ in range(2):
  tr.train_step(batch)
import torch  # noqa: E402
torch.cuda.synchronize()
print('done')
