// Colour augmentation of the camera frame on the device: the imgaug pipeline team_code/data.py:1141-1157 builds
// (image_augmenter: Sequential(random_order=True) of Sometimes(prob, op) over GaussianBlur, AdditiveGaussianNoise, Dropout, Multiply,
// LinearContrast, Grayscale, ElasticTransformation [+ Cutout]) applied to the uint8 frame AFTER the upload, on the prefetcher's copy stream,
// instead of on the loader's CPU workers (data.py:481-496).  The host samples each image's program -- which operators fire, in which order,
// with which parameters (carla_garage_amd/augment.py, the distributions of data.py:1141-1150) -- and this file executes one stage of every
// image's program per launch; pixels are uint8 between stages exactly as between imgaug augmenters (round + saturate after every operator).
// The per-pixel random maps (noise, dropout masks, displacement fields) come from a counter-based generator keyed by (seed, image, stage,
// pixel, channel): imgaug's numpy stream cannot be matched sample by sample, the distributions are (tests/test_augment_gpu.py; the
// operator arithmetic itself is restated in oracle/imgaug_port.py from imgaug 0.4.0 / OpenCV 4.6, requirements.txt:48,95).
// HBM-bound byte work: one thread per pixel (3 channel planes), 9.4 MB per stage at bs = 12.
#include "common.h"
#include "../../include/tfpp.h"

namespace {

__device__ __forceinline__ float u01(unsigned long long key, unsigned long long ctr) {  // (0, 1)
  return ((float)(hash_u32(key * 0x100000001B3ull + ctr) >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float gauss(unsigned long long key, unsigned long long ctr) {  // N(0, 1), Box-Muller on two counters
  const float u1 = u01(key, 2 * ctr), u2 = u01(key ^ 0xA5A5A5A5DEADBEEFull, 2 * ctr + 1);
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
}
__device__ __forceinline__ unsigned char sat_rint(float v) {  // cv2 saturate_cast<uchar>(cvRound(v)) / np.clip(np.round(v)): half to even
  v = rintf(v);
  return (unsigned char)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
}
// products and sums rounded one by one, as numpy / OpenCV's scalar float code rounds them: hipcc contracts a * b + c into one FMA otherwise
// (and __fmul_rn / __fadd_rn are plain operators in the HIP headers), which moves results that sit on a rounding tie
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  const float r = a * b;
  return r;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  const float r = a + b;
  return r;
}
__device__ __forceinline__ int reflect101(int i, int n) {  // cv2.BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
__device__ __forceinline__ float cubic_w(float x) {  // OpenCV INTER_CUBIC (a = -0.75)
  const float a = -0.75f;
  x = fabsf(x);
  if (x <= 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a;
  return 0.f;
}

__global__ __launch_bounds__(256) void aug_stage_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                        const tfpp_aug_op* __restrict__ progs, int stage, int H, int W, unsigned long long seed) {
  const int b = blockIdx.y;
  const long HW = (long)H * W;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= HW) return;
  const tfpp_aug_op op = progs[(size_t)b * TFPP_AUG_MAX_OPS + stage];
  const unsigned char* s = src + (size_t)b * 3 * HW;
  unsigned char* d = dst + (size_t)b * 3 * HW;
  const int y = (int)(pix / W), x = (int)(pix - (long)y * W);
  const unsigned long long key = seed * 0x9E3779B97F4A7C15ull + ((unsigned long long)b << 8) + (unsigned long long)stage;
  switch (op.kind) {
    case TFPP_AUG_NOISE: {  // imgaug AddElementwise on uint8: clip(v + round(N(0, scale)))  (arithmetic.py add_elementwise)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float n = gauss(key, (unsigned long long)(op.per_channel ? c : 0) * HW + pix) * op.a[0];
        d[c * HW + pix] = sat_rint((float)s[c * HW + pix] + rintf(n));
      }
      break;
    }
    case TFPP_AUG_DROPOUT: {  // MultiplyElementwise(Binomial(1 - p)): the pixel (or the channel value) is zeroed with probability p
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bool drop = u01(key, (unsigned long long)(op.per_channel ? c : 0) * HW + pix) < op.a[0];
        d[c * HW + pix] = drop ? 0 : s[c * HW + pix];
      }
      break;
    }
    case TFPP_AUG_MULTIPLY: {  // table[v] = clip(round(v * m))  (arithmetic.py multiply_scalar, uint8 look-up table)
#pragma unroll
      for (int c = 0; c < 3; ++c) {  // (the table is built in float64: the product of an 8-bit value and a float is exact there)
        double t = rint((double)s[c * HW + pix] * (double)op.a[op.per_channel ? c : 0]);
        d[c * HW + pix] = (unsigned char)(t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t));
      }
      break;
    }
    case TFPP_AUG_CONTRAST: {  // table[v] = clip(127 + alpha (v - 127)).astype(uint8): truncation  (contrast.py adjust_contrast_linear)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float t = add_rn(127.f, mul_rn(op.a[op.per_channel ? c : 0], (float)s[c * HW + pix] - 127.f));  // float32 table, no FMA contraction
        t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
        d[c * HW + pix] = (unsigned char)t;
      }
      break;
    }
    case TFPP_AUG_GRAYSCALE: {  // cv2 RGB2GRAY fixed point (4899, 9617, 1868 >> 14), then cv2.addWeighted(gray, alpha, image, 1 - alpha)
      const int r = s[pix], g = s[HW + pix], bl = s[2 * HW + pix];
      const float gray = (float)((r * 4899 + g * 9617 + bl * 1868 + 8192) >> 14);
      const float al = op.a[0];
      const float be = 1.f - al, ga = mul_rn(gray, al);  // products and sum rounded separately (no FMA contraction), as in the float32 host code
      d[pix] = sat_rint(add_rn(ga, mul_rn((float)r, be)));
      d[HW + pix] = sat_rint(add_rn(ga, mul_rn((float)g, be)));
      d[2 * HW + pix] = sat_rint(add_rn(ga, mul_rn((float)bl, be)));
      break;
    }
    case TFPP_AUG_BLUR: {  // cv2.GaussianBlur 5 x 5 (imgaug picks ksize 5 for sigma <= 1.5), BORDER_REFLECT_101; a[0..2] = w(0), w(1), w(2)
      const float w[5] = {op.a[2], op.a[1], op.a[0], op.a[1], op.a[2]};
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy) {
        const int yy = reflect101(y + dy, H);
        float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const long q = (long)yy * W + reflect101(x + dx, W);
#pragma unroll
          for (int c = 0; c < 3; ++c) row[c] += w[dx + 2] * (float)s[c * HW + q];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += w[dy + 2] * row[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c * HW + pix] = sat_rint(acc[c]);
      break;
    }
    case TFPP_AUG_ELASTIC: {  // displacement = alpha * gaussian_blur(U(-1, 1) field, sigma); cv2.remap INTER_CUBIC, constant border 0
      // a[0] = alpha, a[1..3] = 1-D weights w(0), w(1), w(2) of the field's smoothing kernel (sigma = 0.25: w(1) / w(0) = 3.4e-4)
      const float w[5] = {op.a[3], op.a[2], op.a[1], op.a[2], op.a[3]};
      float fx = 0.f, fy = 0.f;
#pragma unroll
      for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const unsigned long long q = (unsigned long long)yy * W + xx;
          const float ww = w[dy + 2] * w[dx + 2];
          fx += ww * (2.f * u01(key, q) - 1.f);
          fy += ww * (2.f * u01(key ^ 0x5555AAAA33331111ull, q) - 1.f);
        }
      const float sx = (float)x + op.a[0] * fx, sy = (float)y + op.a[0] * fy;
      const int x0 = (int)floorf(sx), y0 = (int)floorf(sy);
      const float tx = sx - (float)x0, ty = sy - (float)y0;
      float wx[4], wy[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { wx[k] = cubic_w(tx - (float)(k - 1)); wy[k] = cubic_w(ty - (float)(k - 1)); }
      float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yy = y0 - 1 + j;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int xx = x0 - 1 + i;
          if (xx < 0 || xx >= W) continue;
          const float ww = wy[j] * wx[i];
          const long q = (long)yy * W + xx;
#pragma unroll
          for (int c = 0; c < 3; ++c) acc[c] += ww * (float)s[c * HW + q];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c * HW + pix] = sat_rint(acc[c]);
      break;
    }
    case TFPP_AUG_CUTOUT: {  // imgaug Cutout(squared=False): a = (x1, y1, x2, y2) in pixels, filled with the constant per_channel (cval 128 / 0)
      const bool in = (float)x >= op.a[0] && (float)x < op.a[2] && (float)y >= op.a[1] && (float)y < op.a[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c * HW + pix] = in ? (unsigned char)op.per_channel : s[c * HW + pix];
      break;
    }
    default: {  // this image's program is shorter than the batch's longest one
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c * HW + pix] = s[c * HW + pix];
    }
  }
}

}  // namespace

extern "C" int tfpp_image_augment_stage(const void* src, void* dst, const tfpp_aug_op* progs_dev, int stage, int B, int H, int W, uint64_t seed,
                                        void* stream) {
  if (!src || !dst || src == dst || !progs_dev || stage < 0 || stage >= TFPP_AUG_MAX_OPS || B < 1 || H < 3 || W < 3) return TFPP_EINVAL;
  const long HW = (long)H * W;
  hipLaunchKernelGGL(aug_stage_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)B), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src,
                     (unsigned char*)dst, progs_dev, stage, H, W, (unsigned long long)seed);
  TFPP_CHECK_LAUNCH();
  return 0;
}
