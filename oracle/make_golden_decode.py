"""Writes tests/golden/centernet_decode.npz from the REAL reference (build container only): ``LidarCenterNetHead.get_bboxes``
(team_code/center_net.py:142-170) and ``LidarCenterNet.convert_features_to_bb_metric`` (team_code/model.py:447-459) run
unmodified on the deterministic maps of oracle/decode_port.make_maps.  Usage: python -m oracle.make_golden_decode"""
import os
import types

import numpy as np
import torch

from oracle import decode_port, ref_harness


def main():
  ref_config, ref_model = ref_harness.reference_modules()
  import center_net as ref_cn  # pylint: disable=import-error
  cfg = ref_config.GlobalConfig()
  head = ref_cn.LidarCenterNetHead(cfg)
  out = {}
  for name, batch, seed, peaks in (('b2', 2, 1, 140), ('few', 1, 2, 30)):
    maps = decode_port.make_maps(batch, seed, peaks)
    with torch.no_grad():
      ref = head.get_bboxes(*[m.clone() for m in maps], None, None)
      port = decode_port.decode_heatmap(*maps)
    assert ref.shape == (batch, cfg.top_k_center_keypoints, 9) and ref.dtype == torch.float32
    npos = min(peaks, cfg.top_k_center_keypoints)  # entries beyond the real peaks are background pixels: still distinct scores
    assert torch.equal(ref, port), f'restatement differs from the reference on {name}: {(ref - port).abs().max()}'
    stub = types.SimpleNamespace(head=head, config=cfg)
    with torch.no_grad():
      carla = ref_model.LidarCenterNet.convert_features_to_bb_metric(stub, [m.clone() for m in maps] + [None, None])
    mine = decode_port.convert_features_to_bb_metric(maps)
    assert len(carla) == len(mine) and all(np.array_equal(a, b) for a, b in zip(carla, mine))
    out[f'{name}.boxes'] = ref.numpy()
    out[f'{name}.carla'] = np.stack(carla) if carla else np.zeros((0, 9), np.float32)
    out[f'{name}.args'] = np.array([batch, seed, peaks], dtype=np.int64)
    print(name, 'boxes', tuple(ref.shape), 'above threshold', len(carla), 'npos', npos)
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'centernet_decode.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
