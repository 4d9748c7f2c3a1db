"""Writes tests/golden/lidar_hist.npz from the REAL reference (build container only):
``CARLA_Data.lidar_to_histogram_features`` (team_code/data.py:873-906) run unmodified on the deterministic clouds of
oracle/lidar_port.make_cloud.  Stored as uint8 hit levels (output * hist_max_per_pixel, exact) to keep the fixture small; the
clouds are regenerated from the seed on the GPU box.  Usage: python -m oracle.make_golden_lidar"""
import os
import types

import numpy as np

from oracle import lidar_port, ref_harness


def main():
  ref_config, _ = ref_harness.reference_modules()
  import data as ref_data  # pylint: disable=import-error
  cfg = ref_config.GlobalConfig()
  self_stub = types.SimpleNamespace(config=cfg)
  out = {}
  for name, n, seed in (('sweep60k', 60000, 1), ('sweep5k', 5000, 2), ('empty', 0, 3)):
    cloud = lidar_port.make_cloud(n, seed, edge_cases=n > 0)
    for gp in (False, True):
      ref = ref_data.CARLA_Data.lidar_to_histogram_features(self_stub, cloud.copy(), use_ground_plane=gp)
      assert ref.dtype == np.float32 and ref.shape == ((2 if gp else 1), 256, 256)
      port = lidar_port.lidar_to_histogram_features(cloud, gp)
      assert np.array_equal(ref, port), f'restatement differs from the reference on {name} gp={gp}'
      lvl = np.rint(ref.astype(np.float64) * cfg.hist_max_per_pixel).astype(np.uint8)
      assert np.array_equal((lvl.astype(np.float64) / cfg.hist_max_per_pixel).astype(np.float32), ref)
      out[f'{name}.gp{int(gp)}'] = lvl
    out[f'{name}.n'] = np.int64(cloud.shape[0])
    out[f'{name}.seed'] = np.int64(seed)
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'lidar_hist.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes;', {k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})


if __name__ == '__main__':
  main()
