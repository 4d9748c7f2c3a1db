"""ORACLE (test infrastructure only): rotated-rectangle IoU in EXACT rational arithmetic and the reference's greedy NMS loop.

team_code/transfuser_utils.py:409-450 builds two shapely polygons per pair (``rect_polygon``: corners (+-w, +-h) rotated by ``angle``
radians about the centre, translated to (x, y)) and takes intersection.area / union.area.  shapely is not installed here and does not travel,
so PARITY WITH SHAPELY IS UNPINNED; what this file pins instead is the mathematical quantity shapely approximates: given the float64 corner
coordinates (computed with the same float64 cos / sin / multiply / add as shapely.affinity), the intersection of the two convex quadrilaterals
is clipped with ``fractions.Fraction`` -- no rounding anywhere -- and the IoU is rounded to float64 once at the end.  shapely (GEOS, float64
overlay) agrees with that to ~1e-15 on non-degenerate input.  ``nms_reference`` restates the reference's loop (np.argsort on the
confidences, take the last index, drop every remaining box whose IoU with it exceeds the threshold) on indices."""
import math
from fractions import Fraction

import numpy as np


def rect_corners(x, y, width, height, angle):
  """float64 corners (4, 2) of transfuser_utils.py:434-442 (half extents; shapely.affinity.rotate about the centre, then translate)."""
  c, s = math.cos(angle), math.sin(angle)
  out = []
  for px, py in ((-width, -height), (width, -height), (width, height), (-width, height)):
    out.append((c * px - s * py + x, s * px + c * py + y))
  return np.array(out, dtype=np.float64)


def _area2(poly):
  """twice the signed area, exact"""
  n = len(poly)
  return sum(poly[i][0] * poly[(i + 1) % n][1] - poly[i][1] * poly[(i + 1) % n][0] for i in range(n))


def _clip(subject, clip):
  """Sutherland-Hodgman in Fractions; ``clip`` counter-clockwise."""
  out = list(subject)
  n = len(clip)
  for i in range(n):
    a, b = clip[i], clip[(i + 1) % n]
    ex, ey = b[0] - a[0], b[1] - a[1]
    inp, out = out, []
    if not inp:
      break
    for j, cur in enumerate(inp):
      prev = inp[j - 1]
      sc = ex * (cur[1] - a[1]) - ey * (cur[0] - a[0])
      sp = ex * (prev[1] - a[1]) - ey * (prev[0] - a[0])
      if sc >= 0:
        if sp < 0:
          t = sp / (sp - sc)
          out.append((prev[0] + t * (cur[0] - prev[0]), prev[1] + t * (cur[1] - prev[1])))
        out.append(cur)
      elif sp >= 0:
        t = sp / (sp - sc)
        out.append((prev[0] + t * (cur[0] - prev[0]), prev[1] + t * (cur[1] - prev[1])))
  return out


def iou_exact(bb1, bb2):
  pa = [(Fraction(float(px)), Fraction(float(py))) for px, py in rect_corners(*[float(v) for v in bb1[:5]])]
  pb = [(Fraction(float(px)), Fraction(float(py))) for px, py in rect_corners(*[float(v) for v in bb2[:5]])]
  if _area2(pa) < 0:
    pa = pa[::-1]
  if _area2(pb) < 0:
    pb = pb[::-1]
  inter = _clip(pa, pb)
  ai = abs(_area2(inter)) / 2 if len(inter) >= 3 else Fraction(0)
  union = abs(_area2(pa)) / 2 + abs(_area2(pb)) / 2 - ai
  return float(ai / union) if union > 0 else 0.0


def nms_reference(boxes, thr, iou=iou_exact):
  """Indices kept by transfuser_utils.py:409-431, most confident first (boxes: (N, >= 6) array, confidence in the last column)."""
  boxes = np.asarray(boxes)
  if boxes.size == 0:
    return []
  order = list(np.argsort(boxes[:, -1]))
  kept = []
  while order:
    cur = order.pop()
    kept.append(int(cur))
    order = [j for j in order if iou(boxes[cur], boxes[j]) <= thr]
  return kept


def make_boxes(n, seed, spread=12.0):
  """Clustered vehicle-sized boxes in metres, (n, 9): x, y, half width, half height, yaw, velocity, brake, class, confidence (distinct)."""
  rng = np.random.RandomState(seed)
  centres = rng.uniform(-spread, spread, (max(2, n // 4), 2))
  b = np.zeros((n, 9), np.float32)
  for i in range(n):
    c = centres[rng.randint(len(centres))]
    b[i, 0:2] = c + rng.normal(0, 0.8, 2)
    b[i, 2] = rng.uniform(0.6, 1.3)
    b[i, 3] = rng.uniform(1.5, 2.8)
    b[i, 4] = rng.uniform(-math.pi, math.pi)
    b[i, 7] = rng.randint(0, 4)
  b[:, 8] = rng.permutation(n).astype(np.float32) / n * 0.69 + 0.3
  b[::7, 2] *= -1.0  # the raw wh regression can go negative
  return b
