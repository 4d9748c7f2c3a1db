"""Writes tests/golden/centernet_targets.npz from the REAL reference (build container only): ``CARLA_Data.get_targets``
(team_code/data.py:697-790) run unmodified on the deterministic boxes of oracle/targets_port.make_boxes; asserts that the restatement
reproduces it (integer maps and float64-derived scalars exactly, the float32 gaussian heat-map exactly as well: same numpy).  The boxes are
regenerated from the seed on the GPU box; the fixture holds the reference's maps.  Usage: python -m oracle.make_golden_targets"""
import os
import types

import numpy as np

from oracle import ref_harness, targets_port

CASES = (('many', 30, 1), ('few', 8, 2), ('one', 1, 3), ('none', 0, 4), ('crowd', 64, 5))


def main():
  ref_config, _ = ref_harness.reference_modules()
  import data as ref_data  # pylint: disable=import-error
  cfg = ref_config.GlobalConfig()
  self_stub = types.SimpleNamespace(config=cfg)
  fh = cfg.lidar_resolution_height // cfg.bev_down_sample_factor
  fw = cfg.lidar_resolution_width // cfg.bev_down_sample_factor
  out = {}
  for name, n, seed in CASES:
    boxes = targets_port.make_boxes(n, seed)
    arg = np.array(list(boxes)) if n else np.array([])  # data.py:570: np.array(list of rows); empty list -> shape (0,)
    ref, ref_avg = ref_data.CARLA_Data.get_targets(self_stub, arg, fh, fw)
    port, port_avg = targets_port.get_targets(boxes)
    assert int(ref_avg) == int(port_avg), (name, ref_avg, port_avg)
    for k, v in ref.items():
      assert v.dtype == port[k].dtype and v.shape == port[k].shape, (name, k, v.dtype, v.shape, port[k].dtype, port[k].shape)
      assert np.array_equal(v, port[k]), f'restatement differs from the reference on {name}.{k}'
      out[f'{name}.{k}'] = v
    out[f'{name}.avg_factor'] = np.int64(ref_avg)
    out[f'{name}.n'] = np.int64(n)
    out[f'{name}.seed'] = np.int64(seed)
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'centernet_targets.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
