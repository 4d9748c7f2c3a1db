"""Colour augmentation of the camera frame on the device.

team_code/data.py:63 builds ``image_augmenter(config.color_aug_prob, cutout=config.use_cutout)`` (data.py:1141-1157: imgaug
``Sequential([Sometimes(prob, op) ...], random_order=True)``) and calls it per sample on the decoded HWC uint8 frame inside the DataLoader
workers (data.py:481-496).  Here the frame is uploaded as it was decoded and the same pipeline runs on the GPU, on the prefetcher's copy
stream (``DeviceBatchPrefetcher(..., augment=ImageAugmenter(...))``):

* the HOST samples each image's program with the distributions of data.py:1142-1149 -- per operator a Bernoulli(prob) "fires", a uniform
  random order, and the operator's parameters (blur sigma U(0, 1); noise scale U(0, 12.75); dropout p U(0.01, 0.1); multiply / contrast factor
  U(1/1.2, 1.2); grayscale alpha U(0, 0.5); elastic alpha U(0.5, 1.5), sigma 0.25; ``per_channel=0.5``: a coin per image) -- a few dozen
  numbers per batch;
* the DEVICE executes the programs stage by stage (include/tfpp.h ``tfpp_image_augment_stage``), every image its own operator per stage,
  uint8 between stages like between imgaug augmenters.

There is no CPU fallback: ``apply`` needs a CUDA tensor and the HIP library."""
import ctypes

import numpy as np
import torch

from . import ops
from ._lib import lib

NONE, BLUR, NOISE, DROPOUT, MULTIPLY, CONTRAST, GRAYSCALE, ELASTIC, CUTOUT = range(9)
MAX_OPS = 8
NAMES = ('none', 'blur', 'noise', 'dropout', 'multiply', 'contrast', 'grayscale', 'elastic', 'cutout')


class AugOp(ctypes.Structure):
  _fields_ = [('kind', ctypes.c_int), ('per_channel', ctypes.c_int), ('a', ctypes.c_float * 4)]


OP_DTYPE = np.dtype([('kind', np.int32), ('per_channel', np.int32), ('a', np.float32, (4,))])
assert OP_DTYPE.itemsize == ctypes.sizeof(AugOp) == 24


def gaussian_weights(sigma, half=2):
  """w(0), w(1), w(2) of the normalised 5-tap Gaussian (cv2.getGaussianKernel for sigma > 0)."""
  x = np.arange(-half, half + 1, dtype=np.float64)
  k = np.exp(-(x * x) / (2.0 * float(sigma) ** 2))
  k /= k.sum()
  return k[half:]


class ImageAugmenter:
  """image_augmenter(prob, cutout) of team_code/data.py:1141-1157 with the sampling on the host and the pixels on the device."""

  def __init__(self, prob=0.2, cutout=False, seed=0, cutout_cval=128):
    self.prob, self.cutout, self.cutout_cval = float(prob), bool(cutout), int(cutout_cval)
    self.rng = np.random.default_rng(seed)
    self.calls = 0
    self.seed = int(seed)
    self.kinds = [BLUR, NOISE, DROPOUT, MULTIPLY, CONTRAST, GRAYSCALE, ELASTIC] + ([CUTOUT] if self.cutout else [])
    self._tmp = {}

  # ------------------------------------------------------------------------------------------------ host: programs
  def _params(self, kind, h, w):
    r = self.rng
    op = np.zeros((), OP_DTYPE)
    op['kind'] = kind
    if kind == BLUR:      # ia.GaussianBlur((0, 1.0)); imgaug returns the image unchanged for sigma < 1e-3
      sigma = r.uniform(0.0, 1.0)
      if sigma < 1e-3:
        op['kind'] = NONE
      else:
        op['a'][:3] = gaussian_weights(sigma)
    elif kind == NOISE:   # ia.AdditiveGaussianNoise(loc=0, scale=(0., 0.05 * 255), per_channel=0.5)
      op['a'][0] = r.uniform(0.0, 0.05 * 255)
      op['per_channel'] = int(r.random() < 0.5)
    elif kind == DROPOUT:  # ia.Dropout((0.01, 0.1), per_channel=0.5)
      op['a'][0] = r.uniform(0.01, 0.1)
      op['per_channel'] = int(r.random() < 0.5)
    elif kind in (MULTIPLY, CONTRAST):  # ia.Multiply((1 / 1.2, 1.2), per_channel=0.5), ia.LinearContrast((1 / 1.2, 1.2), per_channel=0.5)
      op['per_channel'] = int(r.random() < 0.5)
      v = r.uniform(1 / 1.2, 1.2, size=3 if op['per_channel'] else 1)
      op['a'][:3] = np.broadcast_to(v, (3,))
    elif kind == GRAYSCALE:  # ia.Grayscale((0.0, 0.5))
      op['a'][0] = r.uniform(0.0, 0.5)
    elif kind == ELASTIC:  # ia.ElasticTransformation(alpha=(0.5, 1.5), sigma=0.25)
      op['a'][0] = r.uniform(0.5, 1.5)
      op['a'][1:4] = gaussian_weights(0.25)
    elif kind == CUTOUT:  # ia.arithmetic.Cutout(squared=False): one rectangle of 0.2 x the image size at a uniform position, constant fill
      cx, cy = r.uniform(0.0, 1.0) * w, r.uniform(0.0, 1.0) * h
      hw, hh = 0.5 * 0.2 * w, 0.5 * 0.2 * h
      op['a'][:] = (max(cx - hw, 0.0), max(cy - hh, 0.0), min(cx + hw, w), min(cy + hh, h))
      op['per_channel'] = self.cutout_cval
    return op

  def sample(self, batch, h, w):
    """programs[batch][MAX_OPS] (numpy, OP_DTYPE) and the number of stages the longest program has."""
    progs = np.zeros((batch, MAX_OPS), OP_DTYPE)
    longest = 0
    for b in range(batch):
      order = self.rng.permutation(len(self.kinds))       # Sequential(random_order=True)
      n = 0
      for j in order:
        if self.rng.random() < self.prob:                  # Sometimes(prob, op)
          op = self._params(self.kinds[j], h, w)
          if op['kind'] != NONE:
            progs[b, n] = op
            n += 1
      longest = max(longest, n)
    return progs, longest

  # ------------------------------------------------------------------------------------------------ device
  def run(self, rgb, progs, stages, seed):
    """Execute `stages` stages of `progs` on rgb (B, 3, H, W) uint8 CUDA; returns the tensor that holds the result (rgb itself or the
    ping-pong buffer of the same shape: valid until the next call on the same stream)."""
    if not (rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.dim() == 4 and rgb.shape[1] == 3 and rgb.is_contiguous()):
      raise ValueError('ImageAugmenter: (B, 3, H, W) contiguous uint8 CUDA frames (the loader\'s layout, data.py:516)')
    if stages == 0:
      return rgb
    b, _, h, w = rgb.shape
    key = (str(rgb.device), torch.cuda.current_stream(rgb.device).cuda_stream, tuple(rgb.shape))
    tmp = self._tmp.get(key)
    if tmp is None:
      tmp = self._tmp[key] = torch.empty_like(rgb)
    dev_progs = torch.from_numpy(progs.view(np.uint8).reshape(-1)).to(rgb.device, non_blocking=True)
    src, dst = rgb, tmp
    for s in range(stages):
      lib.tfpp_image_augment_stage(ops.ptr(src), ops.ptr(dst), ops.ptr(dev_progs), s, b, h, w, int(seed), ops.stream())
      src, dst = dst, src
    return src

  def apply(self, rgb):
    """Sample a program per image and run it (on the current stream)."""
    b, _, h, w = rgb.shape
    progs, stages = self.sample(b, h, w)
    self.calls += 1
    self.last_programs = progs
    return self.run(rgb, progs, stages, (self.seed << 20) + self.calls)
