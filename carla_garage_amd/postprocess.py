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


# ------------------------------------------------------------------------------------------------ on the device
def nms_rotated_device(boxes, iou_threshold, min_conf=float('-inf'), return_iou=False):
  """tfpp_nms_rotated on a (N, S >= 6) float32 CUDA tensor whose LAST column is the confidence: (keep [N] int32 row indices, most confident
  first; count [1] int32) device tensors (+ the (N, N) float64 IoU matrix with ``return_iou``) -- no host synchronisation."""
  import torch
  from . import ops
  from ._lib import lib
  if not (boxes.is_cuda and boxes.dtype == torch.float32 and boxes.dim() == 2 and boxes.is_contiguous()):
    raise ValueError('nms_rotated_device: boxes must be a contiguous (N, S) float32 CUDA tensor')
  n, s = boxes.shape
  keep = torch.empty(max(n, 1), device=boxes.device, dtype=torch.int32)
  if n == 0:
    return (keep, ops.zeros(1, torch.int32, boxes.device)) + ((torch.empty((0, 0), device=boxes.device, dtype=torch.float64),) if return_iou else ())
  count = torch.empty(1, device=boxes.device, dtype=torch.int32)
  iou = torch.empty((n, n), device=boxes.device, dtype=torch.float64) if return_iou else None
  lib.tfpp_nms_rotated(ops.ptr(boxes), n, s, s - 1, float(iou_threshold), float(min_conf), ops.ptr(keep), ops.ptr(count), ops.ptr(iou), ops.stream())
  return (keep, count, iou) if return_iou else (keep, count)


def detect_boxes_nms(models, predictions, iou_threshold):
  """The box post-processing of one sensor_agent.py tick (456-493) on the device: for every model of the ensemble decode the CenterNet maps
  (LidarCenterNetHead.get_bboxes), convert the first sample's boxes to vehicle coordinates (convert_features_to_bb_metric), then ONE greedy
  NMS over all of them.  ``predictions``: one ``pred_bounding_box`` tuple per model.  Returns the reference's list of (9,) float32 arrays, most
  confident first, after a single device -> host copy of the kept rows."""
  import torch
  from . import ops
  from ._lib import lib
  from .config import cfg_get
  rows = []
  for m, bb in zip(models, predictions):
    cfg = m.config
    dec = m.head.get_bboxes(bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], bb[6])[0].contiguous()  # (k, 9), scores descending
    met = torch.empty_like(dec)
    lib.tfpp_bb_image_to_metric(ops.ptr(dec), ops.ptr(met), dec.shape[0], float(cfg.pixels_per_meter), float(cfg.min_x), float(cfg.min_y), ops.stream())
    rows.append(met)
  allb = rows[0] if len(rows) == 1 else torch.cat(rows, 0).contiguous()
  thr = float(cfg_get(models[0].config, 'bb_confidence_threshold', 0.3))
  keep, count = nms_rotated_device(allb, iou_threshold, min_conf=thr)
  host = torch.cat([keep.view(-1).float(), count.float()]).cpu()  # (kept indices, count) in one copy ...
  n = int(host[-1])
  idx = host[:n].long()
  out = allb[idx.to(allb.device)].cpu().numpy() if n else []      # ... and the kept rows in a second one
  return [b for b in out]
