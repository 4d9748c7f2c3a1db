"""CenterNet target rasterisation: the oracle restatement against the fixture written by the reference's
``CARLA_Data.get_targets`` (CPU), and the HIP kernel against both (GPU)."""
import os

import numpy as np
import pytest

from oracle import targets_port

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'centernet_targets.npz')
CASES = ('many', 'few', 'one', 'none', 'crowd')
MAPS = ('center_heatmap_target', 'wh_target', 'offset_target', 'yaw_class_target', 'yaw_res_target', 'velocity_target', 'brake_target', 'pixel_weight')


@pytest.mark.parametrize('name', CASES)
def test_oracle_reproduces_reference_fixture(name):
  g = np.load(GOLDEN)
  boxes = targets_port.make_boxes(int(g[f'{name}.n']), int(g[f'{name}.seed']))
  t, avg = targets_port.get_targets(boxes)
  assert avg == int(g[f'{name}.avg_factor'])
  for k in MAPS:
    assert t[k].dtype == g[f'{name}.{k}'].dtype
    assert np.array_equal(t[k], g[f'{name}.{k}']), k


def test_oracle_edge_semantics():
  """later box owns the cell, heat-map keeps the maximum, brake 0.5 rounds to even, yaw -pi wraps into class 6 with residual 0"""
  b = targets_port.make_boxes(8, 2)
  t, avg = targets_port.get_targets(b)
  x, y = int(77.3 * 0.25), int(190.6 * 0.25)
  assert t['center_heatmap_target'][0, y, x] == 1.0 and t['center_heatmap_target'][2, y, x] == 1.0
  assert t['wh_target'][0, y, x] == np.float32(b[4, 2] * 0.25)
  x7, y7 = int(b[7, 0] * 0.25), int(b[7, 1] * 0.25)
  assert t['brake_target'][y7, x7] == 0
  assert t['yaw_class_target'][y7, x7] == 6 and abs(t['yaw_res_target'][0, y7, x7]) < 1e-6
  assert avg == int((t['center_heatmap_target'] == 1).sum()) >= 7


@pytest.mark.gpu
def test_hip_targets_match_reference_fixture_and_oracle():
  import torch
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import rasterise_targets
  cfg = GlobalConfig()
  g = np.load(GOLDEN)
  nmax = 64
  boxes = np.full((len(CASES), nmax, 8), 1e30)  # rows beyond counts[b] must never be read as boxes
  counts = np.zeros(len(CASES), np.int32)
  for i, name in enumerate(CASES):
    n = int(g[f'{name}.n'])
    boxes[i, :n] = targets_port.make_boxes(n, int(g[f'{name}.seed']))
    counts[i] = n
  out = rasterise_targets(torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda(), cfg)
  torch.cuda.synchronize()
  names = dict(center_heatmap_target='center_heatmap_label', wh_target='wh_label', offset_target='offset_label', yaw_class_target='yaw_class_label',
               yaw_res_target='yaw_res_label', velocity_target='velocity_label', brake_target='brake_target_label', pixel_weight='pixel_weight_label')
  for i, name in enumerate(CASES):
    for k, mine in names.items():
      ref = g[f'{name}.{k}']
      got = out[mine][i].cpu().numpy()
      assert got.shape == ref.shape, (name, k, got.shape, ref.shape)
      if k == 'center_heatmap_target':  # float32 exp: 2 ulp of values <= 1; the set of cells equal to 1 (avg_factor) and the zeros are exact
        assert np.abs(got - ref).max() <= 2.5e-7, (name, np.abs(got - ref).max())
        assert np.array_equal(got == 1, ref == 1) and np.array_equal(got == 0, ref == 0)
      else:
        assert np.array_equal(got, ref.astype(got.dtype)), (name, k)
    assert float(out['avg_factor_label'][i]) == float(g[f'{name}.avg_factor'])


@pytest.mark.gpu
def test_hip_targets_random_batches_against_oracle():
  import torch
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import rasterise_targets
  cfg = GlobalConfig()
  B, nmax = 12, 30
  boxes = np.zeros((B, nmax, 8))
  counts = np.array([(7 * i) % (nmax + 1) for i in range(B)], np.int32)
  for i in range(B):
    boxes[i, :counts[i]] = targets_port.make_boxes(int(counts[i]), 100 + i, edge_cases=i % 2 == 0)
  out = rasterise_targets(torch.from_numpy(boxes).cuda(), torch.from_numpy(counts).cuda(), cfg)
  for i in range(B):
    t, avg = targets_port.get_targets(boxes[i, :counts[i]])
    assert np.abs(out['center_heatmap_label'][i].cpu().numpy() - t['center_heatmap_target']).max() <= 2.5e-7
    for k, mine in (('wh_target', 'wh_label'), ('offset_target', 'offset_label'), ('yaw_res_target', 'yaw_res_label'), ('velocity_target', 'velocity_label'),
                    ('pixel_weight', 'pixel_weight_label'), ('yaw_class_target', 'yaw_class_label'), ('brake_target', 'brake_target_label')):
      assert np.array_equal(out[mine][i].cpu().numpy(), t[k].astype(out[mine].cpu().numpy().dtype)), (i, k)
    assert float(out['avg_factor_label'][i]) == float(avg)
