#!/usr/bin/env python
"""Step time of the DROP-IN path: the module used the way team_code/train.py uses the reference's (train.py:476-531,776-784,883-910) --
DistributedDataParallel (1-rank nccl group, broadcast_buffers=False, find_unused_parameters=False), torch.optim.AdamW(amsgrad), forward ->
model.module.compute_loss -> weighted sum -> backward -> optimizer.step -- against carla_garage_amd.trainer.Trainer on the same batch.
  python tools/dropin_step_time.py [--bs 12] [--steps 10]"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--bs', type=int, default=12)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--dtype', default='bf16')
  args = ap.parse_args()
  from bench import synthetic_batch
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.model import LidarCenterNet
  from carla_garage_amd.trainer import Trainer
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
  dev = torch.device('cuda:0')
  torch.cuda.set_device(dev)
  cfg = GlobalConfig(tfpp_dtype=args.dtype)
  torch.manual_seed(0)
  model = LidarCenterNet(cfg).to(dev).train()
  batch = synthetic_batch(args.bs, cfg, dev, 1234)
  inp = {k: batch[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')}
  lab = {k: v for k, v in batch.items() if k.endswith('_label')}
  lab.setdefault('velocity_label', None)
  lab.setdefault('brake_target_label', None)
  ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], output_device=0, broadcast_buffers=False, find_unused_parameters=False)
  opt = torch.optim.AdamW(ddp.parameters(), lr=3e-4, amsgrad=True)
  w = normalized_loss_weights(cfg)

  def step():
    opt.zero_grad(set_to_none=False)
    out = ddp(**inp)
    losses = ddp.module.compute_loss(pred_wp=out[0], pred_target_speed=out[1], pred_checkpoint=out[2], pred_semantic=out[3], pred_bev_semantic=out[4],
                                     pred_depth=out[5], pred_bounding_box=out[6], pred_wp_1=out[8], selected_path=out[9], **lab)
    total = sum(w[k] * v for k, v in losses.items())
    total.backward()
    opt.step()
    return total

  for _ in range(3):
    step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    last = step()
  torch.cuda.synchronize()
  dropin_ms = 1e3 * (time.perf_counter() - t0) / args.steps
  res = {'bs': args.bs, 'dtype': args.dtype, 'dropin_ms_per_step': round(dropin_ms, 2), 'dropin_samples_per_s': round(args.bs / dropin_ms * 1e3, 1),
         'dropin_loss': round(float(last), 4)}
  del ddp, opt
  tr = Trainer(model)
  for _ in range(3):
    tr.train_step(batch)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    tr.train_step(batch)
  torch.cuda.synchronize()
  res['trainer_eager_ms_per_step'] = round(1e3 * (time.perf_counter() - t0) / args.steps, 2)
  print(json.dumps(res))
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
