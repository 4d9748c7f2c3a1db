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
  write_align_golden(ref_data, cfg)


ALIGN_CASES = (  # name, frames (lidar_seq_len), points per sweep, seed, realign_lidar, y_augmentation [m], yaw_augmentation [deg], use_ground_plane
    ('single', 1, 60000, 11, True, 0.0, 0.0, False),
    ('single_aug', 1, 60000, 12, True, 0.73, -11.5, True),
    ('temporal6_realign', 6, 20000, 13, True, -0.4, 7.25, False),
    ('temporal6_as_recorded', 6, 20000, 14, False, 0.9, 3.0, False),
)


def write_align_golden(ref_data, cfg):
  """tests/golden/lidar_align_hist.npz: the loader's per-frame LiDAR path -- CARLA_Data.align (team_code/data.py:840-871) followed by
  lidar_to_histogram_features (data.py:873-906), chained as CARLA_Data.__getitem__ chains them for the current frame (data.py:524-536) and for
  the temporal frames with and without realign_lidar (data.py:538-560) -- run unmodified on float64 sweeps (laspy .xyz) and ego poses of
  oracle/lidar_port.  Stored as uint8 hit levels per frame plus a few aligned points of every frame."""
  self_stub = types.SimpleNamespace(config=cfg)
  out = {}
  for name, frames, n, seed, realign, y_aug, yaw_aug, gp in ALIGN_CASES:
    meas = lidar_port.make_measurements(seed, frames)
    lv, pts = [], []
    for i in range(frames):
      sweep = lidar_port.make_sweep_f64(n, 100 * seed + i)
      target = meas[frames - 1] if realign else meas[i]  # data.py:542-553
      aligned = ref_data.CARLA_Data.align(self_stub, sweep.copy(), meas[i], target, y_augmentation=y_aug, yaw_augmentation=yaw_aug)
      port = lidar_port.align(sweep, meas[i], target, y_aug, yaw_aug)
      assert aligned.dtype == np.float64 and np.array_equal(aligned, port), f'align restatement differs from the reference ({name}, frame {i})'
      ref = ref_data.CARLA_Data.lidar_to_histogram_features(self_stub, aligned, use_ground_plane=gp)
      assert np.array_equal(ref, lidar_port.lidar_to_histogram_features(aligned, gp))
      lvl = np.rint(ref.astype(np.float64) * cfg.hist_max_per_pixel).astype(np.uint8)
      assert np.array_equal((lvl.astype(np.float64) / cfg.hist_max_per_pixel).astype(np.float32), ref)
      lv.append(lvl)
      pts.append(aligned[:64].copy())
    out[name + '.levels'] = np.stack(lv)            # (frames, C, 256, 256)
    out[name + '.aligned_head'] = np.stack(pts)     # (frames, 64, 3) float64
  path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'lidar_align_hist.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
  main()
