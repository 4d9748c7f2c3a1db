"""Shared helpers for parity tests: golden loading, output packing, tolerance checks."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SEM_STRIDE, BEV_STRIDE, DEPTH_STRIDE = 8, 4, 8  # must match oracle/make_golden.py
GRAD_SAMPLES = 16
# north_star: outputs within 1e-3 relative (fp32) of the CPU reference
REL_TOL_FP32 = 1e-3


def load_golden(name):
  return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def to_np(t):
  return t.detach().float().cpu().numpy()


def pack_outputs(out):
  """Same reduction as oracle/make_golden.py::pack_outputs for any 10-tuple of torch tensors."""
  d = {}
  if out[0] is not None:
    d['pred_wp'] = to_np(out[0])
  if out[1] is not None:
    d['pred_target_speed'] = to_np(out[1])
  if out[2] is not None:
    d['pred_checkpoint'] = to_np(out[2])
  sem, bev, dep = to_np(out[3]), to_np(out[4]), to_np(out[5])
  d['pred_semantic_strided'] = sem[:, :, ::SEM_STRIDE, ::SEM_STRIDE]
  d['pred_bev_semantic_strided'] = bev[:, :, ::BEV_STRIDE, ::BEV_STRIDE]
  d['pred_depth_strided'] = dep[:, ::DEPTH_STRIDE, ::DEPTH_STRIDE]
  for name, a in (('pred_semantic', sem), ('pred_bev_semantic', bev), ('pred_depth', dep)):
    d[name + '_sum'] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])
    d[name + '_rowsum'] = a.astype(np.float64).sum(axis=-1).astype(np.float32)
  for i, name in enumerate(('heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res')):
    d['bb_' + name] = to_np(out[6][i])
  return d


def rel_err(a, b):
  """max |a-b| relative to the reference's scale (max |b|): the 'relative fp32' metric of north_star."""
  a = np.asarray(a, np.float64)
  b = np.asarray(b, np.float64)
  return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, tol, what):
  assert a.shape == b.shape, f'{what}: shape {a.shape} vs {b.shape}'
  assert np.isfinite(a).all(), f'{what}: non-finite values'
  e = rel_err(a, b)
  assert e <= tol, f'{what}: rel err {e:.3e} > {tol:.1e}'
  return e


def compare_packed(got, want, tol=REL_TOL_FP32, keys=None):
  errs = {}
  for k, v in want.items():
    if k not in got or (keys is not None and k not in keys):
      continue
    # sums over millions of pixels accumulate differently: scale by the abs-sum entry
    if k.endswith('_sum'):
      e = abs(got[k][0] - v[0]) / (abs(v[1]) + 1e-30)
      assert e <= tol, f'{k}: checksum rel err {e:.3e}'
      errs[k] = e
    else:
      errs[k] = assert_close(got[k], v, tol, k)
  return errs


def sample_idx(n):
  return np.unique(np.linspace(0, n - 1, GRAD_SAMPLES).astype(np.int64))


def full_outputs(out):
  """{name: full-resolution numpy array} of every tensor of the 10-tuple (no striding, no reduction)."""
  d = {}
  for i, name in ((0, 'pred_wp'), (1, 'pred_target_speed'), (2, 'pred_checkpoint'), (3, 'pred_semantic'), (4, 'pred_bev_semantic'), (5, 'pred_depth')):
    if out[i] is not None:
      d[name] = to_np(out[i])
  if out[6] is not None:
    for i, name in enumerate(('heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res', 'velocity', 'brake')):
      if i < len(out[6]) and out[6][i] is not None:
        d['bb_' + name] = to_np(out[6][i])
  return d


def assert_every_element_close(got_out, want_out, tol, what):
  """|got - want| <= tol * max|want| at EVERY element of every output tensor (the dense maps at full resolution: 1.8 M / 0.7 M / 0.26 M values):
  a single wrong pixel anywhere fails, which the strided samples + row sums of pack_outputs cannot promise."""
  got, want = full_outputs(got_out), full_outputs(want_out)
  assert set(got) == set(want), (sorted(got), sorted(want))
  errs = {}
  for k, w in want.items():
    g = got[k]
    assert g.shape == w.shape, f'{what} {k}: {g.shape} vs {w.shape}'
    d = np.abs(g.astype(np.float64) - w.astype(np.float64))
    scale = float(np.abs(w).max()) + 1e-30
    errs[k] = float(d.max() / scale)
    assert errs[k] <= tol, f'{what} {k}: worst element off by {errs[k]:.3e} of the tensor scale at index {np.unravel_index(int(d.argmax()), d.shape)}'
  return errs
