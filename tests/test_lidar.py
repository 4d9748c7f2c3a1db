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


# ------------------------------------------------------------------------------------------------ the loader's LiDAR path (align + histogram)
ALIGN_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'lidar_align_hist.npz')
ALIGN_CASES = (('single', 1, 60000, 11, True, 0.0, 0.0, False), ('single_aug', 1, 60000, 12, True, 0.73, -11.5, True),
               ('temporal6_realign', 6, 20000, 13, True, -0.4, 7.25, False), ('temporal6_as_recorded', 6, 20000, 14, False, 0.9, 3.0, False))


def _align_case(case):
  name, frames, n, seed, realign, y_aug, yaw_aug, gp = case
  g = np.load(ALIGN_GOLDEN)
  meas = L.make_measurements(seed, frames)
  sweeps = [L.make_sweep_f64(n, 100 * seed + i) for i in range(frames)]
  targets = [meas[frames - 1] if realign else meas[i] for i in range(frames)]
  want = (g[name + '.levels'].astype(np.float64) / L.DEFAULTS['hist_max_per_pixel']).astype(np.float32)
  return sweeps, meas, targets, want, g[name + '.aligned_head']


@pytest.mark.parametrize('case', ALIGN_CASES, ids=[c[0] for c in ALIGN_CASES])
def test_oracle_align_and_histogram_match_reference_golden(case):
  """oracle/lidar_port.align + lidar_to_histogram_features against what the reference's own CARLA_Data.align / lidar_to_histogram_features wrote
  (tests/golden/lidar_align_hist.npz, oracle/make_golden_lidar.py): current frame, augmentation, six temporal frames with and without
  realign_lidar (team_code/data.py:524-560)."""
  sweeps, meas, targets, want, head = _align_case(case)
  _, frames, _, _, _, y_aug, yaw_aug, gp = case
  for i in range(frames):
    al = L.align(sweeps[i], meas[i], targets[i], y_aug, yaw_aug)
    assert np.array_equal(al[:64], head[i])
    assert np.array_equal(L.lidar_to_histogram_features(al, gp), want[i])


def test_align_parameters_of_the_product_equal_the_oracle():
  """carla_garage_amd.lidar.align_params (host side of the device path) against the oracle's restatement of data.py:853-868."""
  from carla_garage_amd.lidar import align_params
  meas = L.make_measurements(5, 3)
  for y_aug, yaw_aug in ((0.0, 0.0), (0.6, -9.0)):
    p = align_params(meas[0], meas[2], y_aug, yaw_aug)
    pd, rd, pa, ra = L.align_params(meas[0], meas[2], y_aug, yaw_aug)
    assert np.array_equal(p, np.array([pd[0], pd[1], pd[2], np.cos(rd), np.sin(rd), pa[0], pa[1], pa[2], np.cos(ra), np.sin(ra)]))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ALIGN_CASES, ids=[c[0] for c in ALIGN_CASES])
def test_hip_align_and_histogram_of_a_batch_bit_exact(case):
  """tfpp_lidar_align_histogram: every frame of a case in ONE call, from the raw float64 sweeps -- histograms equal to the reference's bit for
  bit, aligned points equal to numpy's (the kernel evaluates the 3 x 3 product in numpy's order)."""
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.lidar import LidarBatchHistogram, align_params
  sweeps, meas, targets, want, head = _align_case(case)
  _, frames, _, _, _, y_aug, yaw_aug, gp = case
  hist = LidarBatchHistogram(GlobalConfig())
  params = [align_params(meas[i], targets[i], y_aug, yaw_aug) for i in range(frames)]
  got, aligned = hist(sweeps, params, gp, aligned_out=True)
  assert got.shape == want.shape and got.dtype == torch.float32
  assert np.array_equal(got.cpu().numpy(), want)
  aligned = aligned.cpu().numpy()
  off = 0
  for i in range(frames):
    assert np.array_equal(aligned[off:off + 64], head[i]), f'aligned points of frame {i} differ from numpy'
    off += sweeps[i].shape[0]
  # ragged and empty frames in one batch
  rag = [sweeps[0][:1000], np.zeros((0, 3)), sweeps[0][1000:1777]]
  got2 = hist(rag, [params[0]] * 3, gp).cpu().numpy()
  for j, sw in enumerate(rag):
    assert np.array_equal(got2[j], L.lidar_to_histogram_features(L.align(sw, meas[0], targets[0], y_aug, yaw_aug), gp))
