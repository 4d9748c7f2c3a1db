"""Device-side colour augmentation (carla_garage_amd/augment.py + csrc/augment_kernels.hip) against the numpy restatement of the
imgaug operators team_code/data.py:1141-1157 enables (oracle/imgaug_port.py -- PARITY UNPINNED: imgaug 0.4.0 / OpenCV 4.6 are absent here).

CPU part: the host-side program sampler reproduces the distributions of data.py:1142-1150 (Sometimes(prob), random order, parameter ranges,
per_channel coins), the ABI struct mirror matches include/tfpp.h.
GPU part: operators that are deterministic given their parameters are compared EXACTLY (multiply, contrast, grayscale, cutout) or to one
grey level (blur: summation order of the float kernel) with the oracle; the random maps are checked statistically (noise mean / sigma / shared
vs per-channel maps, dropout rate, displacement range and variance of the elastic transformation measured on ramp images, where bicubic
interpolation is exact); multi-stage programs equal the oracle chain; the prefetcher applies it on the uploaded uint8 frame."""
import re

import numpy as np
import pytest
import torch

from carla_garage_amd import augment as A
from oracle import imgaug_port as O


def test_program_sampler_follows_the_reference_distributions():
  aug = A.ImageAugmenter(prob=0.5, seed=3)
  progs, longest = aug.sample(4000, 256, 1024)
  kinds = progs['kind']
  assert longest <= 7 and kinds.max() <= A.ELASTIC and (kinds[:, 7] == A.NONE).all()
  n_active = (kinds != A.NONE).sum(1)
  assert abs(n_active.mean() - 3.5) < 0.1                      # 7 x Sometimes(0.5)
  for k in (A.BLUR, A.NOISE, A.DROPOUT, A.MULTIPLY, A.CONTRAST, A.GRAYSCALE, A.ELASTIC):
    fires = (kinds == k).any(1).mean()
    assert abs(fires - 0.5) < 0.03, (A.NAMES[k], fires)
    first = (kinds[:, 0] == k).sum() / max(1, (n_active > 0).sum())
    assert abs(first - 1 / 7) < 0.03, (A.NAMES[k], first)       # random_order: every operator equally often in front
  for b in range(50):                                           # active operators are packed to the front, each at most once
    act = kinds[b][kinds[b] != A.NONE]
    assert len(set(act.tolist())) == len(act) and (kinds[b][len(act):] == A.NONE).all()
  sel = lambda k: progs[kinds == k]
  n = sel(A.NOISE)
  assert 0 <= n['a'][:, 0].min() and n['a'][:, 0].max() <= 12.75 and abs(n['per_channel'].mean() - 0.5) < 0.05
  d = sel(A.DROPOUT)
  assert 0.01 <= d['a'][:, 0].min() and d['a'][:, 0].max() <= 0.1
  for k in (A.MULTIPLY, A.CONTRAST):
    m = sel(k)
    assert 1 / 1.2 - 1e-6 <= m['a'][:, :3].min() and m['a'][:, :3].max() <= 1.2 + 1e-6
    shared = m[m['per_channel'] == 0]['a']
    assert (shared[:, 0] == shared[:, 1]).all() and (shared[:, 0] == shared[:, 2]).all()
    per = m[m['per_channel'] == 1]['a']
    assert (per[:, 0] != per[:, 1]).mean() > 0.99
  g = sel(A.GRAYSCALE)
  assert 0 <= g['a'][:, 0].min() and g['a'][:, 0].max() <= 0.5
  e = sel(A.ELASTIC)
  assert 0.5 <= e['a'][:, 0].min() and e['a'][:, 0].max() <= 1.5
  assert np.allclose(e['a'][0, 1:4], O.gaussian_weights(0.25))
  bl = sel(A.BLUR)
  assert np.allclose(bl['a'][:, 0] + 2 * bl['a'][:, 1] + 2 * bl['a'][:, 2], 1.0, atol=1e-6)
  # cutout is part of the pipeline only when the reference asks for it (config.use_cutout, data.py:1152-1153)
  pc, _ = A.ImageAugmenter(prob=1.0, cutout=True, seed=1).sample(20, 256, 1024)
  assert (pc['kind'] == A.CUTOUT).any(1).all() and ((pc['kind'] != A.NONE).sum(1) >= 7).all()
  c = pc[pc['kind'] == A.CUTOUT]
  assert ((c['a'][:, 2] - c['a'][:, 0]) <= 0.2 * 1024 + 1e-3).all() and (c['per_channel'] == 128).all()


def test_abi_struct_mirror_matches_the_header():
  import os
  hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(A.__file__))), 'include', 'tfpp.h'), encoding='utf-8').read()
  assert int(re.search(r'#define\s+TFPP_AUG_MAX_OPS\s+(\d+)', hdr).group(1)) == A.MAX_OPS
  enum = re.search(r'enum \{ TFPP_AUG_NONE = 0,(.*?)\};', hdr, re.S).group(0)
  for name, val in (('BLUR', A.BLUR), ('NOISE', A.NOISE), ('DROPOUT', A.DROPOUT), ('MULTIPLY', A.MULTIPLY), ('CONTRAST', A.CONTRAST),
                    ('GRAYSCALE', A.GRAYSCALE), ('ELASTIC', A.ELASTIC), ('CUTOUT', A.CUTOUT)):
    assert re.search(rf'TFPP_AUG_{name} = {val}\b', enum), name
  assert A.OP_DTYPE.itemsize == 24 and A.OP_DTYPE.fields['a'][1] == 8


def test_oracle_operator_identities():
  """Sanity of the restatement itself: identities every one of these operators must satisfy."""
  rng = np.random.default_rng(0)
  img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
  assert np.array_equal(O.multiply(img, 1.0), img) and np.array_equal(O.linear_contrast(img, 1.0), img)
  assert np.array_equal(O.grayscale(img, 0.0), img)
  g = O.grayscale(img, 1.0)
  assert np.array_equal(g[..., 0], g[..., 1]) and np.array_equal(g[..., 0], g[..., 2])
  assert np.array_equal(O.additive_noise(img, np.zeros((20, 30, 1))), img)
  assert np.array_equal(O.elastic(img, np.zeros((20, 30)), np.zeros((20, 30))), img)       # cubic weights at t = 0 are (0, 1, 0, 0)
  flat = np.full((12, 12, 3), 77, np.uint8)
  assert np.array_equal(O.gaussian_blur(flat, 0.8), flat)                                    # normalised kernel, reflecting border
  assert abs(O.cubic_weights(np.array(0.3)).sum() - 1.0) < 1e-12
  assert O.multiply(np.full((1, 1, 3), 250, np.uint8), 1.2).max() == 255


# ------------------------------------------------------------------------------------------------------------------------------------------ GPU
def _frames(b=3, h=64, w=96, seed=0):
  rng = np.random.default_rng(seed)
  base = rng.integers(0, 256, (b, h // 8, w // 8, 3)).repeat(8, 1).repeat(8, 2)
  img = np.clip(base + rng.integers(-20, 21, (b, h, w, 3)), 0, 255).astype(np.uint8)       # HWC like the decoded frame
  return img


def _dev(img):
  return torch.from_numpy(np.ascontiguousarray(img.transpose(0, 3, 1, 2))).cuda()            # data.py:516: CHW for the network


def _host(t):
  return t.cpu().numpy().transpose(0, 2, 3, 1)


def _run(img, ops_per_image, seed=5):
  aug = A.ImageAugmenter(prob=0.0)
  progs = np.zeros((len(img), A.MAX_OPS), A.OP_DTYPE)
  stages = 0
  for b, lst in enumerate(ops_per_image):
    for s, (kind, pc, a) in enumerate(lst):
      progs[b, s]['kind'], progs[b, s]['per_channel'] = kind, pc
      progs[b, s]['a'][:len(a)] = a
    stages = max(stages, len(lst))
  out = aug.run(_dev(img), progs, stages, seed)
  torch.cuda.synchronize()
  return _host(out)


@pytest.mark.gpu
def test_deterministic_operators_equal_the_oracle():
  img = _frames()
  f32 = lambda *v: tuple(float(np.float32(x)) for x in v)   # the parameters as the device receives them (tfpp_aug_op.a is float)
  m3, a3 = f32(0.85, 1.0, 1.19), f32(1.15, 0.9, 0.84)
  got = _run(img, [[(A.MULTIPLY, 1, m3)], [(A.MULTIPLY, 0, f32(1.2, 1.2, 1.2))], [(A.CONTRAST, 1, a3)]])
  assert np.array_equal(got[0], O.multiply(img[0], m3)) and np.array_equal(got[1], O.multiply(img[1], f32(1.2)[0]))
  assert np.array_equal(got[2], O.linear_contrast(img[2], a3))
  got = _run(img, [[(A.GRAYSCALE, 0, (0.37,))], [(A.GRAYSCALE, 0, (0.5,))], [(A.CUTOUT, 128, (10.3, 5.0, 29.5, 17.8))]])
  assert np.array_equal(got[0], O.grayscale(img[0], 0.37)) and np.array_equal(got[1], O.grayscale(img[1], 0.5))
  assert np.array_equal(got[2], O.cutout(img[2], 10.3, 5.0, 29.5, 17.8, 128))
  for sigma in (0.2, 0.6, 1.0):
    w = O.gaussian_weights(sigma)
    got = _run(img, [[(A.BLUR, 0, w)]] * 3)
    for b in range(3):
      want = O.gaussian_blur(img[b], sigma).astype(np.int32)
      d = np.abs(got[b].astype(np.int32) - want)
      assert d.max() <= 1 and (d > 0).mean() < 0.01, (sigma, d.max(), (d > 0).mean())     # float summation order at exact .5 ties
  # programs of different lengths in one batch, operators in per-image order: the oracle chain, uint8 between the stages
  got = _run(img, [[(A.MULTIPLY, 0, f32(1.1) * 3), (A.GRAYSCALE, 0, (0.2,)), (A.CONTRAST, 1, a3)], [], [(A.CONTRAST, 0, (0.9,) * 3), (A.MULTIPLY, 1, m3)]])
  assert np.array_equal(got[0], O.linear_contrast(O.grayscale(O.multiply(img[0], f32(1.1)[0]), 0.2), a3))
  assert np.array_equal(got[1], img[1])
  assert np.array_equal(got[2], O.multiply(O.linear_contrast(img[2], 0.9), m3))


@pytest.mark.gpu
def test_noise_and_dropout_maps_have_the_reference_statistics():
  h, w = 256, 512
  img = np.full((2, h, w, 3), 128, np.uint8)
  scale = 9.0
  got = _run(img, [[(A.NOISE, 0, (scale,))], [(A.NOISE, 1, (scale,))]]).astype(np.float64) - 128.0
  for b in range(2):
    assert abs(got[b].mean()) < 0.05 and abs(got[b].std() - np.sqrt(scale ** 2 + 1 / 12)) < 0.05, (got[b].mean(), got[b].std())
  assert np.array_equal(got[0][..., 0], got[0][..., 1]) and np.array_equal(got[0][..., 0], got[0][..., 2])  # one map for the three channels
  c = np.corrcoef(got[1][..., 0].ravel(), got[1][..., 1].ravel())[0, 1]
  assert abs(c) < 0.01, c                                                                                      # per_channel: independent maps
  kurt = (got[1] ** 4).mean() / (got[1] ** 2).mean() ** 2
  assert abs(kurt - 3.0) < 0.1, kurt                                                                          # Gaussian, not uniform
  assert np.abs(np.corrcoef(got[1][:, :-1, 0].ravel(), got[1][:, 1:, 0].ravel())[0, 1]) < 0.01                # white
  # saturation like imgaug's clip
  sat = _run(np.full((1, 64, 64, 3), 250, np.uint8), [[(A.NOISE, 0, (12.0,))]])
  assert sat.max() == 255 and (sat == 255).mean() > 0.2
  p = 0.07
  img = _frames(2, h, w, seed=4).clip(1, 255).astype(np.uint8)
  got = _run(img, [[(A.DROPOUT, 0, (p,))], [(A.DROPOUT, 1, (p,))]])
  z0 = got[0] == 0
  assert np.array_equal(z0[..., 0], z0[..., 1]) and np.array_equal(z0[..., 0], z0[..., 2]) and abs(z0[..., 0].mean() - p) < 0.004
  assert np.array_equal(got[0][~z0], img[0][~z0])                                                             # kept pixels untouched
  z1 = got[1] == 0
  assert abs(z1.mean() - p) < 0.004 and abs((z1[..., 0] & z1[..., 1]).mean() - p * p) < 0.002                 # channels dropped independently
  # two calls with different seeds draw different maps, the same seed the same map
  a, b2, c2 = (_run(img[:1], [[(A.DROPOUT, 0, (p,))]], seed=s) for s in (11, 12, 11))
  assert np.array_equal(a, c2) and not np.array_equal(a, b2)


@pytest.mark.gpu
def test_elastic_transformation_displacements_and_remap():
  h, w = 256, 256
  img = _frames(1, h, w, seed=9)
  w25 = O.gaussian_weights(0.25)
  assert np.array_equal(_run(img, [[(A.ELASTIC, 0, (0.0, *w25))]])[0], img[0])                                # no displacement: identity
  ramp_x = np.broadcast_to(np.arange(w, dtype=np.uint8)[None, :, None], (h, w, 3)).copy()
  ramp_y = np.broadcast_to(np.arange(h, dtype=np.uint8)[:, None, None], (h, w, 3)).copy()
  alpha = 1.4
  got = _run(np.stack([ramp_x, ramp_y]), [[(A.ELASTIC, 0, (alpha, *w25))]] * 2, seed=21).astype(np.float64)
  inner = (slice(4, h - 4), slice(4, w - 4))
  dx = (got[0][..., 0] - ramp_x[..., 0])[inner]        # bicubic interpolation reproduces a linear ramp exactly: out = x + dx (rounded)
  dy = (got[1][..., 0] - ramp_y[..., 0])[inner]
  for d in (dx, dy):
    assert np.abs(d).max() <= np.ceil(alpha) and abs(d.mean()) < 0.02
    # U(-alpha, alpha) (the sigma = 0.25 smoothing kernel is a delta to 3e-4) rounded to whole grey levels: P(+-1) = (alpha - 0.5) / (2 alpha)
    assert abs(d.var() - (alpha - 0.5) / alpha) < 0.02, d.var()
  assert np.array_equal(got[0][..., 0], got[0][..., 1])                                     # one displacement field for the three channels
  # against the oracle's remap with the displacement field recovered from the ramps (same seed and stage -> same field for every image)
  tex = _frames(1, h, w, seed=10)
  got_tex = _run(tex, [[(A.ELASTIC, 0, (alpha, *w25))]], seed=21)[0].astype(np.int32)
  gx = _run(ramp_x[None], [[(A.ELASTIC, 0, (alpha, *w25))]], seed=21)[0][..., 0].astype(np.float64) - ramp_x[..., 0]
  # (the recovered field is rounded to whole pixels, so this is a coarse check: most pixels move by the recovered amount +- interpolation)
  assert np.abs(got_tex - tex[0].astype(np.int32)).mean() > 1.0 and np.abs(gx[inner]).max() <= 2   # (at the border the remap reads the constant 0)


@pytest.mark.gpu
def test_prefetcher_augments_the_uploaded_uint8_frame():
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import DeviceBatchPrefetcher
  import test_data_gpu as TD
  cfg = GlobalConfig()
  host = TD._host_batches(cfg, 3)
  for hb in host:
    assert hb['rgb'].dtype == torch.uint8
  plain = [b['rgb'].clone() for b in DeviceBatchPrefetcher(host, cfg)]
  same = [b['rgb'].clone() for b in DeviceBatchPrefetcher(host, cfg, augment=A.ImageAugmenter(prob=0.0))]
  aug = A.ImageAugmenter(prob=1.0, seed=2)
  changed = [b['rgb'].clone() for b in DeviceBatchPrefetcher(host, cfg, augment=aug)]
  torch.cuda.synchronize()
  assert aug.calls == 3
  for p, s, c in zip(plain, same, changed):
    assert p.dtype == torch.float32 and torch.equal(p, s)
    assert c.shape == p.shape and float(c.min()) >= 0 and float(c.max()) <= 255 and float((c - c.round()).abs().max()) == 0
    diff = (c - p).abs()
    assert float(diff.mean()) > 1.0 and float(diff.mean()) < 100.0     # all seven operators fired on a white-noise frame (blur and the remap decorrelate it): another image, valid grey levels
