"""PCIe-inclusive rate of the train step: bs = 12 host batches in the reference's collated layout, (a) uploaded the
reference's way (synchronous .to(device, dtype) from pageable memory, then the step), (b) through DeviceBatchPrefetcher (pinned staging,
copy stream, uint8 frames), both feeding the captured graph step; (c) the resident-input step bench.py reports."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.data import KEYMAP, DeviceBatchPrefetcher, to_reference_batch  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402


def main(bs=12, steps=30, nhost=4):
  cfg = GlobalConfig(tfpp_dtype='bf16')
  dev = torch.device('cuda:0')
  host = [to_reference_batch(bench.synthetic_batch(bs, cfg, None, 1234 + i), cfg) for i in range(nhost)]
  host_bytes_u8 = sum(v.numel() * (1 if k == 'rgb' else (8 if v.dtype == torch.int64 else 4)) for k, v in host[0].items())
  tr = Trainer(LidarCenterNet(cfg).to(dev), lr=1e-5)
  resident = bench.synthetic_batch(bs, cfg, dev, 1234)
  resident = {dst: resident[dst] for src, dst, dt, need in KEYMAP if need(cfg)}
  for _ in range(2):
    tr.train_step(resident)
  step = GraphedTrainStep(tr, resident)

  def ref_upload(b):
    out = {}
    for src, dst, dt, need in KEYMAP:
      if need(cfg):
        t = b[src][:, :cfg.predict_checkpoint_len] if src == 'route' else b[src]
        t = t.to(dev, dtype=dt)
        out[dst] = t.unsqueeze(1) if src == 'speed' else t
    return out

  def timed(gen):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for b in gen:
      step(b)
      n += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

  many = [host[i % nhost] for i in range(steps)]
  res = {'bs': bs, 'host_bytes_per_batch_u8_rgb': host_bytes_u8}
  timed(ref_upload(b) for b in many[:3])
  res['reference_upload_ms'] = timed(ref_upload(b) for b in many)
  timed(DeviceBatchPrefetcher(many[:3], cfg))
  res['prefetcher_ms'] = timed(DeviceBatchPrefetcher(many, cfg))
  res['resident_ms'] = timed(None for _ in many)
  # ... and with the loader's colour augmentation on the device (carla_garage_amd/augment.py: programs sampled on the host, stages on the copy stream)
  from carla_garage_amd.augment import ImageAugmenter
  aug = ImageAugmenter(prob=0.5, seed=1)
  timed(DeviceBatchPrefetcher(many[:3], cfg, augment=aug))
  res['prefetcher_with_augmentation_ms'] = timed(DeviceBatchPrefetcher(many, cfg, augment=aug))
  # ... and with the loader's LiDAR path on the device (round 5): the host batches carry the raw float64 sweeps (60 k points per
  # sample) + align parameters instead of the BEV image; CARLA_Data.align + lidar_to_histogram_features run on the copy stream.  Beside it: what
  # that work costs the loader's workers per batch in numpy (the oracle's restatement of data.py:840-906, one core)
  import numpy as np
  from oracle import lidar_port as LP
  from carla_garage_amd.data import collate_lidar
  from carla_garage_amd.lidar import align_params
  meas = LP.make_measurements(1, 1)
  sweeps = [LP.make_sweep_f64(60000, 500 + j) for j in range(bs)]
  par = [align_params(meas[0], meas[0], 0.3, 5.0) for _ in range(bs)]
  lid_host = []
  for hb in host:
    samples = [{'lidar_sweeps': [sweeps[j]], 'lidar_align': par[j][None]} for j in range(bs)]
    b2 = collate_lidar(samples, cfg)
    b2.update({k: v for k, v in hb.items() if k not in ('lidar', 'temporal_lidar')})
    lid_host.append(b2)
  many_l = [lid_host[i % nhost] for i in range(steps)]
  timed(DeviceBatchPrefetcher(many_l[:3], cfg, lidar_on_device=True))
  res['prefetcher_with_lidar_on_device_ms'] = timed(DeviceBatchPrefetcher(many_l, cfg, lidar_on_device=True))
  res['lidar_raw_bytes_per_batch'] = int(sum(s_.nbytes for s_ in sweeps))
  t0 = time.perf_counter()
  for j in range(bs):
    LP.lidar_to_histogram_features(LP.align(sweeps[j], meas[0], meas[0], 0.3, 5.0), False)
  res['host_numpy_align_histogram_ms_per_batch_one_core'] = (time.perf_counter() - t0) * 1e3
  t0 = time.perf_counter()
  for _ in range(5):
    collate_lidar([{'lidar_sweeps': [sweeps[j]], 'lidar_align': par[j][None]} for j in range(bs)], cfg)
  res['host_collate_raw_sweeps_ms_per_batch'] = (time.perf_counter() - t0) / 5 * 1e3
  # diagnostics: the upload path alone (no step), and the step fed by the prefetcher but replaying on its resident copy
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for b in DeviceBatchPrefetcher(many, cfg):
    pass
  torch.cuda.synchronize()
  res['prefetcher_alone_ms'] = (time.perf_counter() - t0) / len(many) * 1e3
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for b in DeviceBatchPrefetcher(many, cfg):
    step(None)
  torch.cuda.synchronize()
  res['prefetcher_no_d2d_ms'] = (time.perf_counter() - t0) / len(many) * 1e3
  t0 = time.perf_counter()
  pins = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in host[0].items()}
  res['pin_alloc_ms'] = (time.perf_counter() - t0) * 1e3
  t0 = time.perf_counter()
  for b in many[:10]:
    for k, v in b.items():
      pins[k].copy_(v)
  res['host_stage_ms'] = (time.perf_counter() - t0) / 10 * 1e3
  for k in ('reference_upload_ms', 'prefetcher_ms', 'resident_ms'):
    res[k.replace('_ms', '_samples_per_s')] = bs / res[k] * 1e3
  print(json.dumps(res))


if __name__ == '__main__':
  main()
