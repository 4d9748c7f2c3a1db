#!/usr/bin/env python
"""Where does the captured training step spend its time, lane by lane?  TFPP_DEBUG_STAMPS=1 puts a time-stamp launch (device wall clock,
100 MHz) in front of every tape node on the node's own lane, at the lane joins and around the batches of the weight-gradient lane; this
tool replays the captured step, reads the table and prints per lane the time between consecutive stamps grouped by module, the moments the
lanes wait for each other, and the tail after the main chain.  (The ~1400 extra launches stretch the step by ~10 %; the picture is relative.)"""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TFPP_DEBUG_STAMPS'] = '1'
import torch  # noqa: E402

import bench  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402


def group(label):
  m = re.search(r'\[([^\]]+)\]', label)
  key = m.group(1) if m else label.split(' ', 2)[-1]
  for pat, name in ((r'image_encoder\.(stem|s\d)', 'image {}'), (r'lidar_encoder\.(stem|s\d)', 'lidar {}'), (r'transformers\.(\d)', 'gpt {}'),
                    (r'(semantic_decoder|depth_decoder)', '{}'), (r'(join\.layers\.\d)', '{}'), (r'^(head)\.', 'centernet head'),
                    (r'(lidar_channel_to_img|img_channel_to_lidar)\.(\d)', 'token conv'), (r'(up_conv|c5_conv|bev_semantic)', 'bev fpn / sem')):
    mm = re.search(pat, key)
    if mm:
      return name.format(mm.group(1))
  return re.sub(r'[\[\(].*', '', key)[:28]


def main():
  cfg = GlobalConfig(tfpp_dtype='bf16')
  dev = torch.device('cuda:0')
  torch.manual_seed(0)
  model = LidarCenterNet(cfg)
  for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
      torch.nn.init.uniform_(m.weight, 0.5, 1.0)
  tr = Trainer(model.to(dev).train(), lr=1e-5)
  batch = bench.synthetic_batch(12, cfg, dev, 1234)
  step = GraphedTrainStep(tr, batch, warmup=1)
  for _ in range(3):
    step()
  torch.cuda.synchronize()
  st = ops.STAMPS
  t = st['buf'].cpu().numpy()[:st['n']].astype('int64')
  labels = st['labels'][:st['n']]
  t0 = t.min()
  us = (t - t0) / 100.0
  print(f'{len(labels)} stamps, span {us.max() / 1e3:.2f} ms')
  lanes = collections.defaultdict(list)
  for lab, x in zip(labels, us):
    phase, lane = lab.split(' ')[0], lab.split(' ')[1]
    lanes[lane].append((x, phase, lab))
  if '--dump' in sys.argv:  # every stamp of every lane: time since the first stamp, time to the lane's next stamp, label
    with open(sys.argv[sys.argv.index('--dump') + 1], 'w', encoding='utf-8') as f:
      for lane in sorted(lanes):
        ev = sorted(lanes[lane])
        for (x0, ph, lab), (x1, _, _) in zip(ev, ev[1:] + [(ev[-1][0], '', '')]):
          f.write(f'{lane} {x0:10.1f} us  +{x1 - x0:8.1f} us  {lab}\n')
  for lab, x in zip(labels, us):
    if lab.split(' ')[0] == 'step' or 'MAIN CHAIN' in lab or 'joined' in lab:
      print(f'  {x / 1e3:8.3f} ms  {lab}')
  for lane in sorted(lanes):
    ev = sorted(lanes[lane])
    agg = collections.OrderedDict()
    last = {}
    for (x0, ph, lab), (x1, _, _) in zip(ev, ev[1:]):
      g = group(lab)
      if g.startswith('Engine.') and ph in last:
        g = last[ph]  # nodes without a layer key (squeeze-excite, LayerNorm, adds ...) belong to the block around them
      else:
        last[ph] = g
      k = f'{ph} {g}'
      a = agg.setdefault(k, [0.0, 0, x0])
      a[0] += x1 - x0
      a[1] += 1
    print(f'--- {lane}: {len(ev)} stamps, first {ev[0][0] / 1e3:.3f} ms, last {ev[-1][0] / 1e3:.3f} ms')
    for k, (d, n, first) in agg.items():
      if d > 60:
        print(f'   from {first / 1e3:7.3f} ms  {d / 1e3:7.3f} ms in {n:4d} nodes  {k}')
    big = sorted(((x1 - x0, x0, lab) for (x0, _, lab), (x1, _, _) in zip(ev, ev[1:])), reverse=True)[:8]
    print('   longest single gaps:', '; '.join(f'{d:.0f} us at {x0 / 1e3:.2f} ms ({lab.split(" ", 2)[-1][:50]})' for d, x0, lab in big))


if __name__ == '__main__':
  main()
