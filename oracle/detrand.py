"""Deterministic, platform-independent pseudo-random tensors (test infrastructure).

A counter-based generator (splitmix64 finaliser over ``hash(name) + index``)
written in numpy uint64 arithmetic, so the same weights / inputs are produced in
the build container, on the GPU box and under any numpy/torch version -- golden
fixtures (tests/golden) depend on that.  Not part of the product.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
  x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
  z = x
  z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
  z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
  return z ^ (z >> np.uint64(31))


def uniform01(name, shape, seed=0):
  """float64 uniform [0,1) array of ``shape`` keyed by (name, seed)."""
  n = int(np.prod(shape)) if len(shape) else 1
  with np.errstate(over='ignore'):
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) * np.uint64(0x100000001B3) + np.uint64(seed) * np.uint64(
        0x9E3779B1)
    idx = np.arange(n, dtype=np.uint64) + _splitmix64(key)
    bits = _splitmix64(idx)
  u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
  return u.reshape(shape)


def uniform(name, shape, lo=-1.0, hi=1.0, seed=0):
  return (lo + (hi - lo) * uniform01(name, shape, seed)).astype(np.float32)


def randint(name, shape, lo, hi, seed=0):
  """integers in [lo, hi)"""
  return (lo + np.floor(uniform01(name, shape, seed) * (hi - lo))).astype(np.int64)
