"""Host logic of carla_garage_amd/data.py (no GPU): key / dtype mapping of team_code/train.py:688-766 and the box collation."""
import numpy as np
import pytest
import torch

from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.data import KEYMAP, TARGET_KEYS, collate_boxes, to_reference_batch


def test_keymap_follows_train_py():
  cfg = GlobalConfig()
  m = {src: (dst, dt) for src, dst, dt, need in KEYMAP if need(cfg)}
  assert m['rgb'] == ('rgb', torch.float32) and m['lidar'] == ('lidar_bev', torch.float32) and 'temporal_lidar' not in m
  assert m['speed'] == ('ego_vel', torch.float32) and m['route'] == ('checkpoint_label', torch.float32)
  assert m['target_speed'][1] == m['semantic'][1] == m['bev_semantic'][1] == m['yaw_class'][1] == torch.int64  # train.py:699,734,756,760: torch.long
  assert 'velocity' not in m and 'brake_target' not in m and 'ego_waypoints' not in m  # single frame, checkpoint head: no such loss
  t = GlobalConfig(lidar_seq_len=6, lidar_architecture='video_swin_tiny')
  mt = {src: dst for src, dst, dt, need in KEYMAP if need(t)}
  assert mt['temporal_lidar'] == 'lidar_bev' and 'lidar' not in mt and 'velocity' in mt and 'brake_target' in mt
  assert {dst for _, dst, _, _ in KEYMAP} >= set(TARGET_KEYS)


def test_to_reference_batch_host_dtypes():
  import bench
  cfg = GlobalConfig()
  b = to_reference_batch(bench.synthetic_batch(2, cfg, None, 5), cfg)
  assert b['rgb'].dtype == b['semantic'].dtype == b['bev_semantic'].dtype == torch.uint8  # decoded images: data.py:511-522
  assert b['yaw_class'].dtype == torch.int32 and b['speed'].shape == (2,) and b['route'].shape[1] >= cfg.predict_checkpoint_len


def test_collate_boxes():
  lists = [np.arange(16.0).reshape(2, 8), np.array([]), np.ones((5, 8))]
  boxes, counts = collate_boxes(lists)
  assert boxes.dtype == torch.float64 and boxes.shape == (3, 5, 8) and counts.tolist() == [2, 0, 5] and counts.dtype == torch.int32
  assert torch.equal(boxes[0, :2], torch.arange(16.0, dtype=torch.float64).reshape(2, 8)) and float(boxes[0, 2:].abs().sum()) == 0
  assert collate_boxes([np.array([])])[0].shape == (1, 1, 8)
  with pytest.raises(ValueError):
    collate_boxes(lists, max_boxes=4)
