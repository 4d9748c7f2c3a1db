"""TEST INFRASTRUCTURE (oracle): the arithmetic of the imgaug operators team_code/data.py:1141-1157 enables, restated in numpy for GIVEN
parameters and GIVEN random maps, on uint8 HWC images.  Only tests/ may import this file.

PARITY UNPINNED: imgaug 0.4.0 and opencv-python 4.6.0.66 (team_code/requirements.txt:48,95) are third-party dependencies that are absent
from /root/reference and from this image, and there is no network.  What follows restates their published algorithms:
  * imgaug/augmenters/arithmetic.py (0.4.0): add_elementwise (uint8: image + round(value), clipped), multiply_scalar (uint8: 256-entry
    look-up table clip(round(arange(256) * m))), MultiplyElementwise with a Binomial(1 - p) mask (Dropout), Cutout;
  * imgaug/augmenters/contrast.py adjust_contrast_linear (uint8: table = 127 + alpha (arange(256) - 127), clipped, .astype(uint8));
  * imgaug/augmenters/color.py Grayscale -> cv2.cvtColor(RGB2GRAY) (8-bit fixed point 4899 / 9617 / 1868, shift 14) blended by
    imgaug/augmenters/blend.py blend_alpha (uint8, scalar alpha: cv2.addWeighted(fg, alpha, bg, 1 - alpha, 0) = saturate(round()));
  * imgaug/augmenters/blur.py blur_gaussian_ (cv2 backend: ksize = max(3.3 sigma, 5) for sigma < 3, odd; BORDER_REFLECT_101; identity for
    sigma < 1e-3) with OpenCV's float Gaussian kernel exp(-x^2 / (2 sigma^2)) normalised to 1 (OpenCV's 8-bit path quantises the kernel to
    fixed point: results may differ from this float restatement by 1 grey level);
  * imgaug/augmenters/geometric.py ElasticTransformation (displacement = alpha * gaussian-smoothed U(-1, 1) field, cv2.remap INTER_CUBIC,
    BORDER_CONSTANT 0; OpenCV quantises the coordinates to 1/32 pixel: a 1-2 grey level difference to this float restatement).
The anchor for parity is the reference's own call site (data.py:481-496: the augmenter is called per sample on the HWC uint8 frame before the
CHW transpose) and these formulas; every function below names the one it restates."""
import numpy as np


def _sat(v):
  return np.clip(np.rint(v), 0, 255).astype(np.uint8)  # np.round / cvRound: half to even


def additive_noise(img, noise):
  """arithmetic.add_elementwise, uint8: noise is the float N(0, scale) map, (H, W, 3) or (H, W, 1)."""
  return np.clip(img.astype(np.int32) + np.rint(noise).astype(np.int32), 0, 255).astype(np.uint8)


def dropout(img, keep):
  """arithmetic.MultiplyElementwise with a 0/1 mask (Dropout): keep (H, W, 3) or (H, W, 1)."""
  return (img * keep.astype(np.uint8)).astype(np.uint8)


def multiply(img, m):
  """arithmetic.multiply_scalar, uint8 look-up table; m scalar or 3 per-channel factors."""
  m = np.broadcast_to(np.asarray(m, np.float64).reshape(-1), (3,)) if np.ndim(m) else np.full(3, float(m))
  out = np.empty_like(img)
  for c in range(3):
    table = np.clip(np.rint(np.arange(256, dtype=np.float64) * m[c]), 0, 255).astype(np.uint8)
    out[..., c] = table[img[..., c]]
  return out


def linear_contrast(img, alpha):
  """contrast.adjust_contrast_linear, uint8: table = 127 + alpha (v - 127), clipped, truncated by astype."""
  a = np.broadcast_to(np.asarray(alpha, np.float32).reshape(-1), (3,)) if np.ndim(alpha) else np.full(3, np.float32(alpha))
  out = np.empty_like(img)
  v = np.arange(256, dtype=np.float32)
  for c in range(3):
    table = np.clip(np.float32(127) + a[c] * (v - np.float32(127)), 0, 255).astype(np.uint8)
    out[..., c] = table[img[..., c]]
  return out


def grayscale(img, alpha):
  """color.Grayscale: cv2 RGB2GRAY (fixed point) blended with the image by cv2.addWeighted."""
  r, g, b = (img[..., i].astype(np.int64) for i in range(3))
  gray = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.float32)
  al = np.float32(alpha)
  return _sat(gray[..., None] * al + img.astype(np.float32) * (np.float32(1) - al))


def gaussian_weights(sigma, half=2):
  """cv2.getGaussianKernel(2 * half + 1, sigma) for sigma > 0: exp(-x^2 / (2 sigma^2)), normalised; returns w(0..half)."""
  x = np.arange(-half, half + 1, dtype=np.float64)
  k = np.exp(-(x * x) / (2.0 * float(sigma) ** 2))
  k /= k.sum()
  return k[half:]


def gaussian_blur(img, sigma):
  """blur.blur_gaussian_ (cv2 backend), sigma <= 1.5 -> 5 x 5, BORDER_REFLECT_101, float kernel."""
  if sigma < 1e-3:
    return img.copy()
  w = gaussian_weights(sigma)
  k = np.concatenate([w[:0:-1], w]).astype(np.float32)
  pad = np.pad(img.astype(np.float32), ((2, 2), (2, 2), (0, 0)), mode='reflect')  # numpy 'reflect' = REFLECT_101
  h, wd = img.shape[:2]
  rows = sum(k[i] * pad[:, i:i + wd] for i in range(5))
  out = sum(k[i] * rows[i:i + h] for i in range(5))
  return _sat(out)


def cubic_weights(t):
  """OpenCV interpolateCubic (A = -0.75): weights of the taps at -1, 0, 1, 2 for the fractional offset t."""
  a = -0.75
  x = np.abs(np.stack([t + 1, t, 1 - t, 2 - t], -1))
  near = ((a + 2) * x - (a + 3)) * x * x + 1
  far = ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
  return np.where(x <= 1, near, np.where(x < 2, far, 0.0))


def elastic(img, dx, dy):
  """geometric.ElasticTransformation remap: out(y, x) = bicubic(img, x + dx, y + dy), BORDER_CONSTANT 0 (float coordinates)."""
  h, w = img.shape[:2]
  yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
  sx, sy = xx + dx.astype(np.float32), yy + dy.astype(np.float32)
  x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
  wx, wy = cubic_weights(sx - x0), cubic_weights(sy - y0)
  src = np.pad(img.astype(np.float32), ((4, 4), (4, 4), (0, 0)))
  out = np.zeros(img.shape, np.float32)
  for j in range(4):
    for i in range(4):
      ys, xs = np.clip(y0 - 1 + j + 4, 0, h + 7), np.clip(x0 - 1 + i + 4, 0, w + 7)
      out += (wy[..., j] * wx[..., i])[..., None] * src[ys, xs]
  return _sat(out)


def cutout(img, x1, y1, x2, y2, cval):
  """arithmetic.Cutout(squared=False, fill_mode='constant')."""
  out = img.copy()
  out[int(np.ceil(y1)):int(np.ceil(y2)), int(np.ceil(x1)):int(np.ceil(x2))] = cval
  return out
