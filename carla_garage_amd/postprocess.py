"""Host-side post-processing after the CenterNet decode: rotated-box IoU and the greedy non-maximum suppression that merges the boxes of
several models / frames (team_code/transfuser_utils.py:409-450, called from team_code/sensor_agent.py:491 on at most a few dozen boxes
per frame -- host work in the reference too).

The reference builds shapely polygons (``rect_polygon``: half-extents, rotation about the centre, counter-clockwise, radians) and takes
``intersection.area / union.area``.  shapely is not a dependency of this package: the intersection of two convex quadrilaterals is
clipped directly (Sutherland-Hodgman) and the union is area(a) + area(b) - area(a & b), which is what shapely's union area equals."""
import numpy as np


def rect_corners(x, y, width, height, angle):
  """Corners (4, 2), counter-clockwise, of the rectangle of transfuser_utils.py:434-442: centre (x, y), HALF extents (width, height),
  rotated by ``angle`` radians counter-clockwise."""
  c, s = np.cos(angle), np.sin(angle)
  local = np.array([(-width, -height), (width, -height), (width, height), (-width, height)], dtype=np.float64)
  rot = np.array([[c, -s], [s, c]])
  return local @ rot.T + np.array([x, y], dtype=np.float64)


def polygon_area(p):
  if len(p) < 3:
    return 0.0
  x, y = p[:, 0], p[:, 1]
  return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def signed_area(p):
  x, y = p[:, 0], p[:, 1]
  return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def clip_convex(subject, clip):
  """Sutherland-Hodgman: the part of the convex polygon ``subject`` inside the convex, counter-clockwise polygon ``clip``."""
  out = [tuple(p) for p in subject]
  n = len(clip)
  if signed_area(clip) < 0.0:  # a negative half-extent (raw CenterNet wh regression) flips the ring: shapely does not care, the clipper does
    clip = clip[::-1]
  for i in range(n):
    a, b = clip[i], clip[(i + 1) % n]
    edge = b - a
    inp, out = out, []
    if not inp:
      break

    def side(p):
      return edge[0] * (p[1] - a[1]) - edge[1] * (p[0] - a[0])  # > 0: left of a -> b = inside

    for j, cur in enumerate(inp):
      prev = inp[j - 1]
      sc, sp = side(cur), side(prev)
      if sc >= 0:
        if sp < 0:
          t = sp / (sp - sc)
          out.append((prev[0] + t * (cur[0] - prev[0]), prev[1] + t * (cur[1] - prev[1])))
        out.append(cur)
      elif sp >= 0:
        t = sp / (sp - sc)
        out.append((prev[0] + t * (cur[0] - prev[0]), prev[1] + t * (cur[1] - prev[1])))
  return np.array(out, dtype=np.float64).reshape(-1, 2)


def iou_bbs(bb1, bb2):
  """transfuser_utils.py:445-450; boxes are (x, y, half_width, half_height, yaw, ...)."""
  a = rect_corners(float(bb1[0]), float(bb1[1]), float(bb1[2]), float(bb1[3]), float(bb1[4]))
  b = rect_corners(float(bb2[0]), float(bb2[1]), float(bb2[2]), float(bb2[3]), float(bb2[4]))
  inter = polygon_area(clip_convex(a, b))
  union = polygon_area(a) + polygon_area(b) - inter
  return inter / union if union > 0.0 else 0.0


def non_maximum_suppression(bounding_boxes, iou_treshhold):
  """transfuser_utils.py:409-431: ``bounding_boxes`` is a list (one entry per model / frame) of lists of boxes whose LAST element is the
  confidence.  Greedy: repeatedly keep the most confident remaining box and drop every remaining box whose IoU with it exceeds the
  threshold.  Returns the kept boxes, most confident first."""
  boxes = [np.asarray(b) for group in bounding_boxes for b in group]
  if not boxes:
    return []
  # the reference sorts an object array with numpy's default argsort (transfuser_utils.py:411,416): the same call, so equal confidences
  # come out in the same order
  conf = np.empty(len(boxes), dtype=object)
  conf[:] = [b[-1] for b in boxes]
  order = list(np.argsort(conf))
  kept = []
  while order:
    idx = order.pop()
    cur = boxes[idx]
    kept.append(cur)
    order = [j for j in order if iou_bbs(cur, boxes[j]) <= iou_treshhold]
  return kept
