"""Test infrastructure: CPU restatement of the CenterNet heat-map decode and box conversion.

Follows ``LidarCenterNetHead.decode_heatmap`` / ``class2angle`` (team_code/center_net.py:125-140,172-237), the helpers
``get_local_maximum`` / ``get_topk_from_heatmap`` / ``transpose_and_gather_feat`` (team_code/gaussian_target.py:186-264),
``LidarCenterNet.convert_features_to_bb_metric`` (team_code/model.py:447-459) and ``bb_image_to_vehicle_system``
(team_code/transfuser_utils.py:388-406) -- the step right after ``forward`` on the 20 Hz path (team_code/sensor_agent.py:463-467).
Plain PyTorch on CPU, fp32, operation by operation in the reference's order (so the float roundings match).  Pinned against the
reference's own functions by oracle/make_golden_decode.py -> tests/golden/centernet_decode.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import detrand

DEFAULTS = dict(lidar_resolution_height=256, lidar_resolution_width=256, top_k_center_keypoints=100, center_net_max_pooling_kernel=3,
                num_dir_bins=12, num_bb_classes=4, bb_confidence_threshold=0.3, pixels_per_meter=4.0, min_x=-32, min_y=-32)


def decode_heatmap(heat, wh, offset, yaw_class, yaw_res, cfg=None):
  """(B, k, 9): x, y, w, h (image pixels), yaw, velocity (0), brake (0), class, score -- single-frame configuration."""
  c = dict(DEFAULTS, **(cfg or {}))
  k, kernel = c['top_k_center_keypoints'], c['center_net_max_pooling_kernel']
  B, _, fh, fw = heat.shape
  height_ratio, width_ratio = float(c['lidar_resolution_height'] / fh), float(c['lidar_resolution_width'] / fw)
  hmax = F.max_pool2d(heat, kernel, stride=1, padding=(kernel - 1) // 2)
  heat = heat * (hmax == heat).float()
  scores, inds = torch.topk(heat.reshape(B, -1), k)
  clses = torch.div(inds, fh * fw, rounding_mode='trunc')
  inds = inds % (fh * fw)
  ys = torch.div(inds, fw, rounding_mode='trunc')
  xs = (inds % fw).int().float()

  def gather(feat):
    feat = feat.permute(0, 2, 3, 1).contiguous().view(B, -1, feat.shape[1])
    return feat.gather(1, inds.unsqueeze(2).repeat(1, 1, feat.shape[2]))

  whg, og, ycg, yrg = gather(wh), gather(offset), gather(yaw_class), gather(yaw_res)
  ycls = torch.argmax(ycg, -1)
  angle = ycls.float() * (2 * np.pi / float(c['num_dir_bins'])) + yrg.squeeze(2)
  angle[angle > np.pi] -= 2 * np.pi
  zeros = torch.zeros_like(angle)
  xs = xs + og[..., 0]
  ys = ys + og[..., 1]
  boxes = torch.stack([xs, ys, whg[..., 0], whg[..., 1], angle, zeros, zeros], dim=2)
  boxes = torch.cat((boxes, clses[..., None], scores[..., None]), dim=-1)
  boxes[:, :, 0] *= width_ratio
  boxes[:, :, 1] *= height_ratio
  boxes[:, :, 2] *= width_ratio
  boxes[:, :, 3] *= height_ratio
  return boxes


def bb_image_to_vehicle_system(box, pixels_per_meter, min_x, min_y):
  box = box.copy()
  box[4] = -box[4]
  translation = np.array([-(min_x * pixels_per_meter), -(min_y * pixels_per_meter)])
  box[:2] = box[:2] - translation
  box[0], box[1] = box[1], box[0]
  box[2], box[3] = box[3], box[2]
  box[:4] = box[:4] / pixels_per_meter
  return box


def convert_features_to_bb_metric(maps, cfg=None):
  """list of (9,) float32 arrays for the first sample of the batch, as team_code/model.py:447-459 returns it."""
  c = dict(DEFAULTS, **(cfg or {}))
  boxes = decode_heatmap(*maps[:5], cfg=cfg)[0]
  boxes = boxes[boxes[:, -1] > c['bb_confidence_threshold']]
  return [bb_image_to_vehicle_system(b, c['pixels_per_meter'], c['min_x'], c['min_y']) for b in boxes.detach().cpu().numpy()]


def make_maps(batch, seed=0, peaks=140, cfg=None):
  """Deterministic CenterNet head outputs: a heat-map (after sigmoid) with ``peaks`` isolated local maxima of distinct heights
  per sample on a low, strictly-varying background, and random regression maps.  All scores are distinct, so top-k has no ties."""
  c = dict(DEFAULTS, **(cfg or {}))
  ncls, nb, H, W = c['num_bb_classes'], c['num_dir_bins'], 64, 64
  n = ncls * H * W
  base = detrand.uniform01(f'dec.base.{seed}', (batch, n)) * 0.05                       # background in (0, 0.05)
  heat = base.copy()
  for b in range(batch):
    pos = np.argsort(detrand.uniform01(f'dec.pos.{seed}.{b}', (n,)))[:peaks]
    heat[b, pos] = 0.1 + 0.89 * (np.arange(peaks)[::-1] + detrand.uniform01(f'dec.h.{seed}.{b}', (peaks,)) * 0.5) / peaks
  heat = torch.from_numpy(heat.reshape(batch, ncls, H, W).astype(np.float32))
  u = lambda name, ch, lo, hi: torch.from_numpy(detrand.uniform(f'dec.{name}.{seed}', (batch, ch, H, W), lo, hi))
  return heat, u('wh', 2, 0.5, 12.0), u('off', 2, 0.0, 1.0), u('yc', nb, -3.0, 3.0), u('yr', 1, -0.3, 0.3)
