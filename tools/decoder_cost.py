#!/usr/bin/env python
"""What do the six fp32 planning-decoder layers cost per training step?  Step time of the captured bs = 12 bf16 step with
num_transformer_decoder_layers = 6 (default) / 3 / 1 (timing only: another model)."""
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402


def main():
  dev = torch.device('cuda:0')
  for rep in range(2):
    for layers in (6, 3, 1):
      cfg = GlobalConfig(tfpp_dtype='bf16', num_transformer_decoder_layers=layers)
      torch.manual_seed(0)
      tr = Trainer(LidarCenterNet(cfg).to(dev).train(), lr=1e-5)
      batch = bench.synthetic_batch(12, cfg, dev, 1234)
      step = GraphedTrainStep(tr, batch, warmup=1)
      for _ in range(3):
        step()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(20):
        step()
      torch.cuda.synchronize()
      print(f'decoder layers {layers}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step', flush=True)
      del step, tr
      torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
