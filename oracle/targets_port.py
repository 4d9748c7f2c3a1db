"""Test infrastructure: CPU restatement of the CenterNet target rasterisation of the training data path.

Follows ``CARLA_Data.get_targets`` (team_code/data.py:697-790) with ``gaussian_radius`` / ``gen_gaussian_target`` / ``gaussian2d``
(team_code/gaussian_target.py:11-61,64-187) and ``angle2class`` (team_code/center_net.py:240-254).  The reference calls it with the
UNPADDED float64 box array built by ``parse_bounding_boxes`` (data.py:565-591: rows = x, y, extent_x, extent_y, yaw, speed, brake, class in
BEV image pixels), so every scalar below is float64 arithmetic written into float32 / int32 maps; only the gaussian patch is float32.
Pinned against the reference itself by oracle/make_golden_targets.py -> tests/golden/centernet_targets.npz.
"""
import math

import numpy as np

from oracle import detrand

DEFAULTS = dict(lidar_resolution_height=256, lidar_resolution_width=256, bev_down_sample_factor=4, num_bb_classes=4, num_dir_bins=12)
MIN_OVERLAP = 0.1  # data.py:753


def gaussian_radius(height, width, min_overlap):
  """gaussian_target.py:166-187 (three quadratic cases, smallest root)"""
  b1 = height + width
  c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
  r1 = (b1 - math.sqrt(b1 * b1 - 4 * c1)) / 2
  b2 = 2 * (height + width)
  c2 = (1 - min_overlap) * width * height
  r2 = (b2 - math.sqrt(b2 * b2 - 16 * c2)) / 8
  a3 = 4 * min_overlap
  b3 = -2 * min_overlap * (height + width)
  c3 = (min_overlap - 1) * width * height
  r3 = (b3 + math.sqrt(b3 * b3 - 4 * a3 * c3)) / (2 * a3)
  return min(r1, r2, r3)


def gaussian_patch(radius):
  """gaussian_target.py:11-30 at sigma = (2 r + 1) / 6, float32"""
  sigma = (2 * radius + 1) / 6
  ax = np.arange(-radius, radius + 1, dtype=np.float32)
  h = np.exp(-(ax[None, :] * ax[None, :] + ax[:, None] * ax[:, None]) / (2 * sigma * sigma))
  h[h < np.finfo(np.float32).eps * h.max()] = 0
  return h


def angle2class(angle, num_dir_bins):
  """center_net.py:240-254"""
  angle = angle % (2 * np.pi)
  per = 2 * np.pi / float(num_dir_bins)
  shifted = (angle + per / 2) % (2 * np.pi)
  cls = shifted // per
  return int(cls), shifted - (cls * per + per / 2)


def get_targets(boxes, cfg=None):
  """boxes: (n, 8) float64 -> (dict of maps, avg_factor), names and dtypes of data.py:722-729,781-790"""
  c = dict(DEFAULTS, **(cfg or {}))
  fh = c['lidar_resolution_height'] // c['bev_down_sample_factor']
  fw = c['lidar_resolution_width'] // c['bev_down_sample_factor']
  wr, hr = float(fw / c['lidar_resolution_width']), float(fh / c['lidar_resolution_height'])
  t = dict(center_heatmap_target=np.zeros((c['num_bb_classes'], fh, fw), np.float32), wh_target=np.zeros((2, fh, fw), np.float32),
           offset_target=np.zeros((2, fh, fw), np.float32), yaw_class_target=np.zeros((fh, fw), np.int32),
           yaw_res_target=np.zeros((1, fh, fw), np.float32), velocity_target=np.zeros((1, fh, fw), np.float32),
           brake_target=np.zeros((fh, fw), np.int32), pixel_weight=np.zeros((2, fh, fw), np.float32))
  boxes = np.asarray(boxes, dtype=np.float64)
  if boxes.ndim < 2 or boxes.shape[0] == 0:
    return t, 1
  for b in boxes:
    ctx, cty = b[0] * wr, b[1] * hr
    x, y = int(ctx), int(cty)  # astype(int): toward zero
    ex, ey = b[2] * wr, b[3] * hr
    radius = max(2, int(gaussian_radius(ey, ex, MIN_OVERLAP)))
    heat = t['center_heatmap_target'][int(b[7])]
    g = gaussian_patch(radius)
    left, right = min(x, radius), min(fw - x, radius + 1)
    top, bottom = min(y, radius), min(fh - y, radius + 1)
    region = heat[y - top:y + bottom, x - left:x + right]
    np.maximum(region, g[radius - top:radius + bottom, radius - left:radius + right], out=region)
    t['wh_target'][:, y, x] = (ex, ey)
    t['yaw_class_target'][y, x], t['yaw_res_target'][0, y, x] = angle2class(b[4], c['num_dir_bins'])
    t['velocity_target'][0, y, x] = b[5]
    t['brake_target'][y, x] = int(round(b[6]))
    t['offset_target'][:, y, x] = (ctx - x, cty - y)
    t['pixel_weight'][:, y, x] = 1.0
  return t, max(1, int(np.equal(t['center_heatmap_target'], 1).sum()))


def make_boxes(n, seed, edge_cases=True):
  """Deterministic (n, 8) float64 boxes in BEV image pixels (what parse_bounding_boxes + bb_vehicle_to_image_system emit: centres strictly
  inside the 256 x 256 grid).  With edge_cases the first rows sit at the grid borders, share a centre cell, and carry extreme extents."""
  u = detrand.uniform01('centernet_boxes', (max(n, 1), 8), seed)[:n]
  b = np.zeros((n, 8), np.float64)
  b[:, 0] = 0.5 + u[:, 0] * 255.0
  b[:, 1] = 0.5 + u[:, 1] * 255.0
  b[:, 2] = 0.3 + u[:, 2] * 30.0
  b[:, 3] = 0.3 + u[:, 3] * 12.0
  b[:, 4] = (u[:, 4] - 0.5) * 2 * np.pi
  b[:, 5] = u[:, 5] * 20.0
  b[:, 6] = u[:, 6]
  b[:, 7] = np.floor(u[:, 7] * 4).clip(0, 3)
  if edge_cases and n >= 8:
    b[0, :2] = (0.2, 0.3)          # top-left cell, patch clipped on two sides
    b[1, :2] = (255.9, 255.7)      # bottom-right cell
    b[2, :2] = (128.0, 4.0)        # exactly on a cell corner: offset 0
    b[3, :2] = b[4, :2] = (77.3, 190.6)  # two boxes in one cell: the later one owns the scalar targets
    b[3, 7], b[4, 7] = 0, 2
    b[5, 2:4] = (120.0, 90.0)      # huge box: radius > 20
    b[6, 2:4] = (0.01, 0.01)       # tiny box: radius floor of 2
    b[7, 4] = -np.pi               # yaw on the wrap-around
    b[7, 6] = 0.5                  # brake on the rounding split: round-half-even -> 0
  return b
