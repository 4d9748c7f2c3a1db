"""Test infrastructure: CPU restatement of the LiDAR -> BEV histogram.

Follows ``CARLA_Data.lidar_to_histogram_features`` (team_code/data.py:873-906), the step immediately before
``LidarCenterNet.forward`` on the 20 Hz path (team_code/sensor_agent.py:421-425): points above ``max_height_lidar`` are
dropped, the rest are split at ``lidar_split_height`` and each part is binned on a fixed (max_x-min_x)*ppm x
(max_y-min_y)*ppm grid with ``np.histogramdd`` semantics (bin i = [e_i, e_i+1), the last bin closed on the right),
counts clipped at ``hist_max_per_pixel`` and divided by it, x/y transposed, channels = [below, above] or [above].
Pinned against the reference itself by oracle/make_golden_lidar.py -> tests/golden/lidar_hist.npz.
"""
import numpy as np

from oracle import detrand

DEFAULTS = dict(min_x=-32, max_x=32, min_y=-32, max_y=32, pixels_per_meter=4.0, hist_max_per_pixel=5, lidar_split_height=0.2,
                max_height_lidar=100.0)


def bin_edges(lo, hi, ppm):
  """exactly the edges the reference builds (team_code/data.py:883-886)"""
  return np.linspace(lo, hi, (hi - lo) * int(ppm) + 1)


def bin_counts(points, lo, hi, ppm):
  """np.histogramdd's bin index along one axis (numpy/lib/_histograms_impl.py: searchsorted(edges, x, 'right'), values equal to
  the last edge moved into the last bin, everything outside -- NaN included -- dropped): -1 = outside."""
  edges = bin_edges(lo, hi, ppm)
  n = len(edges) - 1
  v = points.astype(np.float64)
  idx = np.searchsorted(edges, v, side='right') - 1
  idx[v == edges[-1]] = n - 1
  idx[(idx < 0) | (idx >= n)] = -1
  return idx, n


def lidar_to_counts(lidar, use_ground_plane, cfg=None):
  """Integer hit counts per cell *before* clipping, shape (C, ny, nx) (already transposed like the reference output)."""
  c = dict(DEFAULTS, **(cfg or {}))
  lidar = np.asarray(lidar)
  lidar = lidar[lidar[..., 2] < c['max_height_lidar']]
  parts = [lidar[lidar[..., 2] <= c['lidar_split_height']], lidar[lidar[..., 2] > c['lidar_split_height']]]
  if not use_ground_plane:
    parts = parts[1:]
  out = []
  for pts in parts:
    ix, nx = bin_counts(pts[:, 0], c['min_x'], c['max_x'], c['pixels_per_meter'])
    iy, ny = bin_counts(pts[:, 1], c['min_y'], c['max_y'], c['pixels_per_meter'])
    ok = (ix >= 0) & (iy >= 0)
    hist = np.zeros((nx, ny), dtype=np.int64)
    np.add.at(hist, (ix[ok], iy[ok]), 1)
    out.append(hist.T)
  return np.stack(out, axis=0)


def lidar_to_histogram_features(lidar, use_ground_plane, cfg=None):
  """(C, H, W) float32 exactly as team_code/data.py:873-906 returns it."""
  c = dict(DEFAULTS, **(cfg or {}))
  counts = lidar_to_counts(lidar, use_ground_plane, cfg).astype(np.float64)
  counts[counts > c['hist_max_per_pixel']] = c['hist_max_per_pixel']
  return (counts / c['hist_max_per_pixel']).astype(np.float32)


def make_cloud(n, seed=0, edge_cases=True):
  """Deterministic synthetic sweep: n points in a 80 m x 80 m x 8 m box (part of it outside the grid), float32, plus points
  exactly on cell edges, on the outer borders, at the split height and at the height cut-off."""
  xyz = np.stack([detrand.uniform(f'lidar.x.{seed}', (n,), -40.0, 40.0), detrand.uniform(f'lidar.y.{seed}', (n,), -40.0, 40.0),
                  detrand.uniform(f'lidar.z.{seed}', (n,), -3.0, 5.0)], axis=1)
  # a dense patch so that the clip at hist_max_per_pixel is exercised
  k = n // 10
  xyz[:k, 0] = detrand.uniform(f'lidar.dx.{seed}', (k,), 3.0, 6.0)
  xyz[:k, 1] = detrand.uniform(f'lidar.dy.{seed}', (k,), -2.0, 2.0)
  if edge_cases:
    e = np.array([[-32.0, -32.0, 1.0], [32.0, 32.0, 1.0], [32.0, -32.0, 1.0], [-32.0, 32.0, 1.0], [0.0, 0.0, 1.0], [0.25, -0.25, 1.0],
                  [31.75, 31.75, 1.0], [np.nextafter(np.float32(32.0), np.float32(0)), 0.0, 1.0], [np.nextafter(np.float32(32.0), np.float32(64)), 0.0, 1.0],
                  [np.nextafter(np.float32(-32.0), np.float32(-64)), 0.0, 1.0], [-1e-7, 1e-7, 1.0], [1.0, 1.0, 0.2], [1.0, 1.0, np.nextafter(np.float32(0.2), np.float32(1))],
                  [2.0, 2.0, 100.0], [2.0, 2.0, np.nextafter(np.float32(100.0), np.float32(0))], [5.125, -7.375, -2.0]], dtype=np.float32)
    xyz = np.concatenate([xyz, e], axis=0)
  return np.ascontiguousarray(xyz.astype(np.float32))


# ----------------------------------------------------------------------------------------------------------------------------------
# The loader's per-frame LiDAR path: CARLA_Data.align + lidar_to_histogram_features as
# CARLA_Data.__getitem__ chains them (team_code/data.py:524-560), restated in numpy.
# ----------------------------------------------------------------------------------------------------------------------------------
def normalize_angle(x):
  """team_code/transfuser_utils.py:19-23"""
  x = x % (2 * np.pi)
  if x > np.pi:
    x -= 2 * np.pi
  return x


def algin_lidar(lidar, translation, yaw):
  """team_code/transfuser_utils.py:116-130 (sic): rotation inverse to translation and yaw."""
  rotation_matrix = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
  return (rotation_matrix.T @ (lidar - translation).T).T


def align_params(measurements_0, measurements_1, y_augmentation=0.0, yaw_augmentation=0):
  """The two (translation, yaw) pairs CARLA_Data.align derives (team_code/data.py:853-868) from the measurements of the sweep's frame and
  of the target frame and from the augmentation: returns (pos_diff (3,), rot_diff, pos_diff_aug (3,), rot_diff_aug)."""
  pos_1 = np.array([measurements_1['pos_global'][0], measurements_1['pos_global'][1], 0.0])
  pos_0 = np.array([measurements_0['pos_global'][0], measurements_0['pos_global'][1], 0.0])
  pos_diff = pos_1 - pos_0
  rot_diff = normalize_angle(measurements_1['theta'] - measurements_0['theta'])
  rotation_matrix = np.array([[np.cos(measurements_1['theta']), -np.sin(measurements_1['theta']), 0.0],
                              [np.sin(measurements_1['theta']), np.cos(measurements_1['theta']), 0.0], [0.0, 0.0, 1.0]])
  pos_diff = rotation_matrix.T @ pos_diff
  return pos_diff, rot_diff, np.array([0.0, y_augmentation, 0.0]), np.deg2rad(yaw_augmentation)


def align(lidar_0, measurements_0, measurements_1, y_augmentation=0.0, yaw_augmentation=0):
  """team_code/data.py:840-871"""
  pos_diff, rot_diff, pos_diff_aug, rot_diff_aug = align_params(measurements_0, measurements_1, y_augmentation, yaw_augmentation)
  return algin_lidar(algin_lidar(lidar_0, pos_diff, rot_diff), pos_diff_aug, rot_diff_aug)


def make_sweep_f64(n, seed):
  """A deterministic float64 sweep as laspy's .xyz yields it (team_code/data.py:365): make_cloud's geometry on a 1 mm lattice (the .laz scale),
  no hand-placed edge points (after a rotation they are ordinary points)."""
  c = make_cloud(n, seed, edge_cases=False).astype(np.float64)
  return np.round(c * 1000.0) / 1000.0


def make_measurements(seed, frames):
  """`frames` consecutive ego poses (pos_global, theta) of a car driving a gentle curve at ~8 m/s, 20 Hz frames 0.05 s apart."""
  th0 = float(detrand.uniform(f'meas.th.{seed}', (1,), -3.0, 3.0)[0])
  x, y, out = 100.0 + seed, -50.0 + 2 * seed, []
  for i in range(frames):
    th = th0 + 0.01 * i
    x, y = x + 0.4 * np.cos(th), y + 0.4 * np.sin(th)
    out.append({'pos_global': [float(x), float(y)], 'theta': float(th)})
  return out
