"""Rotated-box IoU / greedy NMS (carla_garage_amd/postprocess.py) -- transfuser_utils.py:409-450.  The reference needs shapely, which is
not installed here, so these are known-answer and property tests: closed-form overlaps, a dense point-sampling estimate of
random rotated pairs, and the suppression order of the reference's loop restated on the IoU matrix."""
import math

import numpy as np

from carla_garage_amd.postprocess import iou_bbs, non_maximum_suppression, rect_corners


def test_iou_known_answers():
  a = [0.0, 0.0, 1.0, 2.0, 0.3]
  assert abs(iou_bbs(a, a) - 1.0) < 1e-12
  assert iou_bbs([0, 0, 1, 1, 0.0], [5, 0, 1, 1, 0.7]) == 0.0
  # axis-aligned 2x2 squares shifted by 1: intersection 2, union 6
  assert abs(iou_bbs([0, 0, 1, 1, 0.0], [1, 0, 1, 1, 0.0]) - 2.0 / 6.0) < 1e-12
  # the same square turned by 45 degrees: the intersection is a regular octagon of area 8 (sqrt 2 - 1)
  oct_area = 8.0 * (math.sqrt(2.0) - 1.0)
  assert abs(iou_bbs([0, 0, 1, 1, 0.0], [0, 0, 1, 1, math.pi / 4]) - oct_area / (8.0 - oct_area)) < 1e-12
  # half extents: width 2 / height 1 box inside a 4 x 4 one -> 8 / 64
  assert abs(iou_bbs([0, 0, 2, 1, 0.0], [0, 0, 4, 4, 0.0]) - 8.0 / 64.0) < 1e-12
  # a quarter turn swaps the extents
  assert abs(iou_bbs([0, 0, 2, 1, math.pi / 2], [0, 0, 1, 2, 0.0]) - 1.0) < 1e-9


def _inside(pts, corners):
  ok = np.ones(len(pts), bool)
  for i in range(4):
    a, b = corners[i], corners[(i + 1) % 4]
    ok &= (b[0] - a[0]) * (pts[:, 1] - a[1]) - (b[1] - a[1]) * (pts[:, 0] - a[0]) >= 0
  return ok


def test_iou_against_point_sampling_and_symmetry():
  rng = np.random.RandomState(0)
  g = np.linspace(-8, 8, 1601)
  pts = np.stack(np.meshgrid(g, g), -1).reshape(-1, 2)
  for _ in range(12):
    b1 = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(-math.pi, math.pi)]
    b2 = [rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(0.5, 2.5), rng.uniform(0.5, 2.5), rng.uniform(-math.pi, math.pi)]
    ia, ib = _inside(pts, rect_corners(*b1)), _inside(pts, rect_corners(*b2))
    est = (ia & ib).sum() / max((ia | ib).sum(), 1)
    got = iou_bbs(b1, b2)
    assert abs(got - est) < 5e-3, (got, est)
    assert abs(got - iou_bbs(b2, b1)) < 1e-12 and 0.0 <= got <= 1.0


def test_nms_follows_the_reference_loop():
  rng = np.random.RandomState(1)
  groups = []
  for _ in range(3):  # three "models", six boxes each, clustered so that many overlap
    groups.append([np.array([rng.normal(0, 2.0), rng.normal(0, 2.0), 1.0 + rng.rand(), 2.0 + rng.rand(), rng.uniform(-0.3, 0.3), 0.0, 0.0, 0.0,
                             rng.rand()]) for _ in range(6)])
  thr = 0.2
  kept = non_maximum_suppression(groups, thr)
  flat = [b for g in groups for b in g]
  conf = np.array([b[-1] for b in flat])
  idx = list(np.argsort(conf))  # the reference's bookkeeping (transfuser_utils.py:417-429) on indices
  want = []
  while idx:
    cur = idx.pop()
    want.append(cur)
    idx = [j for j in idx if iou_bbs(flat[cur], flat[j]) <= thr]
  assert [float(b[-1]) for b in kept] == [float(conf[i]) for i in want]
  assert all(kept[i][-1] >= kept[i + 1][-1] for i in range(len(kept) - 1))
  for i in range(len(kept)):
    for j in range(i + 1, len(kept)):
      assert iou_bbs(kept[i], kept[j]) <= thr
  assert non_maximum_suppression([[], []], thr) == []


def test_iou_with_negative_half_extents_equals_the_same_rectangles():
  """The raw wh regression can be negative: shapely's polygon of (-w, h) covers the same points as that of (w, h) (the ring just runs the
  other way), so the IoU must not change -- the clipper used to return 0 for these."""
  rng = np.random.RandomState(3)
  for _ in range(20):
    b1 = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), rng.uniform(-math.pi, math.pi)]
    b2 = [rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.5, 2.0), rng.uniform(0.5, 2.0), rng.uniform(-math.pi, math.pi)]
    want = iou_bbs(b1, b2)
    assert want > 0.0
    for s1 in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
      for s2 in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
        n1 = [b1[0], b1[1], s1[0] * b1[2], s1[1] * b1[3], b1[4]]
        n2 = [b2[0], b2[1], s2[0] * b2[2], s2[1] * b2[3], b2[4]]
        assert abs(iou_bbs(n1, n2) - want) < 1e-12


def test_host_iou_and_nms_against_the_exact_oracle_fixture():
  """tests/golden/nms.npz (oracle/make_golden_nms.py): IoU matrices computed in exact rational arithmetic from the float64 corners and the
  indices the reference's greedy loop keeps.  The host implementation must reproduce the IoUs to rounding and the kept boxes exactly."""
  import os
  g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'nms.npz')))
  for tag in 'abc':
    b, iou, kept, thr = g[f'boxes_{tag}'], g[f'iou_{tag}'], g[f'kept_{tag}'], float(g[f'thr_{tag}'])
    n = len(b)
    step = 1 if n <= 100 else 7
    for i in range(0, n, step):
      for j in range(i + 1, n, step):
        assert abs(iou_bbs(b[i], b[j]) - iou[i, j]) < 1e-12, (tag, i, j)
    got = non_maximum_suppression([list(b[:n // 2]), list(b[n // 2:])], thr)
    assert [float(x[-1]) for x in got] == [float(b[k, -1]) for k in kept], tag


def test_oracle_exact_iou_equals_closed_forms():
  from oracle import nms_port as N
  assert N.iou_exact([0, 0, 1, 1, 0.0], [1, 0, 1, 1, 0.0]) == 2.0 / 6.0
  assert N.iou_exact([0, 0, 2, 1, 0.0], [0, 0, 4, 4, 0.0]) == 8.0 / 64.0
  assert N.iou_exact([0, 0, 1, 1, 0.0], [5, 0, 1, 1, 0.7]) == 0.0
  oct_area = 8.0 * (math.sqrt(2.0) - 1.0)
  assert abs(N.iou_exact([0, 0, 1, 1, 0.0], [0, 0, 1, 1, math.pi / 4]) - oct_area / (8.0 - oct_area)) < 1e-15
  assert N.iou_exact([0.3, -0.2, -1.2, 2.0, 0.4], [0.3, -0.2, 1.2, 2.0, 0.4]) == 1.0   # a negative half extent is the same rectangle
