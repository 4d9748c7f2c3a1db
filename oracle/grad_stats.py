"""Statistics that compare two sets of gradients of the same network (reduced precision against fp32).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by oracle/make_golden_bf16.py (the reference under autocast against its own fp32
step), tests/test_model.py (the HIP bf16 step against the HIP fp32 step) and bench.py's bf16 leg, so that all three numbers are the
same formula.
"""
import numpy as np
import torch

STAT_KEYS = ('arena_cosine', 'arena_rel_l2', 'norm_err_median', 'norm_err_p90', 'norm_err_p99', 'norm_err_max', 'beyond_half')
GRAD_SAMPLES = 16


def sample_idx(n):
  return np.unique(np.linspace(0, n - 1, GRAD_SAMPLES).astype(np.int64))


def gradient_stats(ref, got):
  """ref, got: {parameter name: gradient tensor}, ``ref`` the fp32 one.  Whole arena: cosine and relative L2 distance.  Per tensor (those
  carrying >= 1e-3 of the largest norm): relative error of the norm -- median / p90 / p99 / max over tensors -- and the fraction of 16 sampled
  elements per tensor that differ by more than 0.5 x (rms + |ref|)."""
  names = [n for n in ref if n in got]
  a = torch.cat([ref[n].flatten().double() for n in names])
  b = torch.cat([got[n].flatten().double() for n in names])
  out = {'arena_cosine': float((a * b).sum() / (a.norm() * b.norm())), 'arena_rel_l2': float((b - a).norm() / a.norm())}
  big = max(float(ref[n].double().norm()) for n in names)
  nerr, beyond, total = [], 0, 0
  for n in names:
    r, g = ref[n].flatten().double(), got[n].flatten().double()
    rn = float(r.norm())
    if rn < 1e-3 * big:
      continue
    nerr.append(abs(float(g.norm()) - rn) / rn)
    idx = torch.from_numpy(sample_idx(r.numel())).to(r.device)
    rel = (g[idx] - r[idx]).abs() / (rn / np.sqrt(r.numel()) + r[idx].abs())
    beyond += int((rel > 0.5).sum())
    total += int(rel.numel())
  e = np.array(nerr)
  out.update(norm_err_median=float(np.median(e)), norm_err_p90=float(np.percentile(e, 90)), norm_err_p99=float(np.percentile(e, 99)),
             norm_err_max=float(e.max()), beyond_half=beyond / max(total, 1), tensors=len(nerr))
  return out
