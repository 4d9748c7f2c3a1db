"""LiDAR -> BEV histogram: oracle vs. the reference's golden output (CPU), HIP path vs. both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import lidar_port as L

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lidar_hist.npz')
CASES = ('sweep60k', 'sweep5k', 'empty')


def _golden(name, gp):
  g = np.load(GOLDEN)
  cloud = L.make_cloud(int(g[f'{name}.n']) - (16 if int(g[f'{name}.n']) else 0), int(g[f'{name}.seed']), edge_cases=int(g[f'{name}.n']) > 0)
  assert cloud.shape[0] == int(g[f'{name}.n'])
  want = (g[f'{name}.gp{int(gp)}'].astype(np.float64) / L.DEFAULTS['hist_max_per_pixel']).astype(np.float32)
  return cloud, want


@pytest.mark.parametrize('gp', [False, True])
@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_golden(name, gp):
  cloud, want = _golden(name, gp)
  got = L.lidar_to_histogram_features(cloud, gp)
  assert got.dtype == np.float32 and got.shape == want.shape
  assert np.array_equal(got, want)
  if name == 'sweep60k':
    assert want.max() == 1.0 and 0.0 < want.mean() < 0.5  # the clip at hist_max_per_pixel is exercised, the grid is sparse


def test_oracle_bin_semantics():
  """np.histogramdd edge rules on the reference grid: left-closed cells, right-closed last cell, outside / NaN dropped."""
  pts = np.array([[-32.0, 0.0, 1], [32.0, 0.0, 1], [np.nextafter(np.float32(32), np.float32(64)), 0.0, 1], [0.0, 0.0, 1], [-1e-7, 0.0, 1],
                  [np.nan, 0.0, 1], [0.0, np.inf, 1], [0.0, 0.0, np.nan]], dtype=np.float32)
  c = L.lidar_to_counts(pts, False)[0]
  assert c.sum() == 4 and c[128, 0] == 1 and c[128, 255] == 1 and c[128, 128] == 1 and c[128, 127] == 1
  ref = np.histogramdd(pts[:5, :2].astype(np.float64), bins=(L.bin_edges(-32, 32, 4.0), L.bin_edges(-32, 32, 4.0)))[0]
  assert np.array_equal(ref.T, c)


@pytest.mark.gpu
@pytest.mark.parametrize('gp', [False, True])
@pytest.mark.parametrize('name', CASES)
def test_hip_histogram_bit_exact(name, gp):
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.lidar import LidarHistogram
  cloud, want = _golden(name, gp)
  hist = LidarHistogram(GlobalConfig())
  got = hist(cloud, gp)
  assert got.shape == want.shape and got.dtype == torch.float32 and got.is_cuda
  assert np.array_equal(got.cpu().numpy(), want)
  assert np.array_equal(got.cpu().numpy(), L.lidar_to_histogram_features(cloud, gp))
  if cloud.shape[0]:  # extra columns (intensity) and a device-resident cloud take the same path
    wide = torch.from_numpy(np.concatenate([cloud, np.ones((cloud.shape[0], 1), np.float32)], axis=1)).cuda()
    assert np.array_equal(hist(wide, gp).cpu().numpy(), want)


@pytest.mark.gpu
def test_hip_histogram_other_grid_and_properties():
  """A non-default grid (edges that are not dyadic rationals) against numpy's own histogramdd, and size-independent properties at
  1 M points: total mass = clipped count of in-range points, permutation invariance."""
  import types
  from carla_garage_amd.lidar import LidarHistogram
  cfg = types.SimpleNamespace(min_x=-30, max_x=30, min_y=-21, max_y=21, pixels_per_meter=3.0, hist_max_per_pixel=7, lidar_split_height=0.2,
                              max_height_lidar=100.0)
  cloud = L.make_cloud(200000, 7)
  hist = LidarHistogram(cfg)
  got = hist(cloud, True).cpu().numpy()
  want = L.lidar_to_histogram_features(cloud, True, vars(cfg))
  assert got.shape == (2, 126, 180) and np.array_equal(got, want)
  big = L.make_cloud(1000000, 9, edge_cases=False)
  h2 = LidarHistogram(types.SimpleNamespace(**L.DEFAULTS))
  a = h2(big, True).cpu().numpy()
  perm = np.random.default_rng(0).permutation(big.shape[0])
  assert np.array_equal(a, h2(big[perm], True).cpu().numpy())
  counts = L.lidar_to_counts(big, True)
  assert int(np.rint(a.astype(np.float64).sum() * 5)) == int(np.minimum(counts, 5).sum())
