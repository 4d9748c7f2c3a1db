"""Writes tests/golden/nms.npz: boxes, the exact IoU matrix (oracle/nms_port.py: rational arithmetic) and the indices the reference's
greedy loop keeps, for three sizes.   python -m oracle.make_golden_nms"""
import os

import numpy as np

from oracle import nms_port as N

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def main():
  out = {}
  for tag, n, seed, thr in (('a', 24, 1, 0.2), ('b', 100, 2, 0.2), ('c', 300, 3, 0.35)):
    b = N.make_boxes(n, seed)
    iou = np.zeros((n, n), np.float64)
    for i in range(n):
      for j in range(i + 1, n):
        iou[i, j] = iou[j, i] = N.iou_exact(b[i], b[j])
    kept = N.nms_reference(b, thr, iou=None) if False else None
    order = list(np.argsort(b[:, -1]))
    kept = []
    while order:
      cur = order.pop()
      kept.append(int(cur))
      order = [j for j in order if iou[cur, j] <= thr]
    out.update({f'boxes_{tag}': b, f'iou_{tag}': iou, f'kept_{tag}': np.array(kept, np.int32), f'thr_{tag}': np.array(thr)})
    print(tag, n, 'kept', len(kept), 'pairs overlapping', int((iou > 0).sum() // 2))
  np.savez_compressed(os.path.join(GOLDEN, 'nms.npz'), **out)


if __name__ == '__main__':
  main()
