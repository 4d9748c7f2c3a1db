// Data-movement kernels of the Video-Swin LiDAR backbone (BASELINE config 5; team_code/video_swin_transformer.py): the 3-D patch
// embedding gather, the row gather that implements zero padding + cyclic shift + window partition (and their inverse, and the 2x2
// PatchMerging concatenation) from a precomputed index table, and the window-attention softmax with relative-position bias and
// shift mask.  All HBM-bound; 16-byte accesses per lane.
#include "common.h"
#include "../../include/tfpp.h"

// x: fp32 (B, T, H, W) frames (the reference views them as (B, 1, T, H, W), transfuser.py:153) -> rows (b, t/2, h/4, w/4) of 32 values
// ordered (kt, kh, kw): the im2col of Conv3d(1, 96, kernel (2, 4, 4), stride (2, 4, 4)) (video_swin_transformer.py:427-467).
template <typename T>
__global__ void patchify3d_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int Tn, int H, int W) {
  const int Do = Tn / 2, Ho = H / 4, Wo = W / 4;
  const long total = (long)B * Do * Ho * Wo * 8;  // one thread per (row, kt, kh): 4 consecutive kw
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int part = (int)(i & 7), kt = part >> 2, kh = part & 3;
  long r = i >> 3;
  const int wo = (int)(r % Wo); r /= Wo;
  const int ho = (int)(r % Ho); r /= Ho;
  const int dd = (int)(r % Do);
  const int b = (int)(r / Do);
  const float4 v = *reinterpret_cast<const float4*>(x + (((size_t)b * Tn + 2 * dd + kt) * H + 4 * ho + kh) * W + 4 * wo);
  T* o = out + (i >> 3) * 32 + part * 4;
  o[0] = ElemTraits<T>::from_f(v.x); o[1] = ElemTraits<T>::from_f(v.y); o[2] = ElemTraits<T>::from_f(v.z); o[3] = ElemTraits<T>::from_f(v.w);
}

extern "C" int tfpp_patchify3d(const float* x, void* out, int B, int T, int H, int W, int dtype, void* stream) {
  if (!x || !out || B < 1 || T < 2 || (T & 1) || (H & 3) || (W & 3)) return TFPP_EINVAL;
  const long total = (long)B * (T / 2) * (H / 4) * (W / 4) * 8;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(patchify3d_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, (float*)out, B, T, H, W);
  else hipLaunchKernelGGL(patchify3d_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out, B, T, H, W);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dst[r][c0 .. c0 + C) = (idx[r] >= 0 ? src[idx[r]][0 .. C) : 0) + (add ? add[r][0 .. C) : 0)
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int* __restrict__ idx, const T* __restrict__ add, T* __restrict__ dst,
                                   long rows, int CV, long src_ld, long dst_ld, long add_ld) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * CV) return;
  const long r = i / CV;
  const int c = (int)(i - r * CV) * VEC;
  const int s = idx[r];
  float v[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] = 0.f;
  if (s >= 0) load_vec<T>(src + (size_t)s * src_ld + c, v);
  if (add) {
    float a[VEC];
    load_vec<T>(add + (size_t)r * add_ld + c, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] += a[e];
  }
  store_vec<T>(dst + (size_t)r * dst_ld + c, v);
}

extern "C" int tfpp_gather_rows(const void* src, const int32_t* idx, const void* add, void* dst, int64_t rows, int C, int64_t src_ld,
                                int64_t dst_ld, int64_t add_ld, int dtype, void* stream) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (!src || !idx || !dst || rows < 1 || C < VEC || C % VEC || src_ld % VEC || dst_ld % VEC || (add && add_ld % VEC)) return TFPP_EINVAL;
  const long n = (long)rows * (C / VEC);
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(gather_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, (const int*)idx, (const float*)add,
                       (float*)dst, (long)rows, C / VEC, (long)src_ld, (long)dst_ld, (long)add_ld);
  else
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (const int*)idx, (const bf16_t*)add,
                       (bf16_t*)dst, (long)rows, C / VEC, (long)src_ld, (long)dst_ld, (long)add_ld);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Window attention scores (video_swin_transformer.py:139-166), in place: S[(w, h), i, :] <- softmax_j(alpha * S + table[rel_index[i, j], h]
// + mask[w % n_mask, i, j]).  One wave per row; n <= 64 * 4.  rel_index: int32 [n, n] (the [:n, :n] corner of the module's buffer),
// table: fp32 [(2Wd-1)(2Wh-1)(2Ww-1), heads], mask: fp32 [n_mask, n, n] of 0 / -100 or NULL.
#define SWIN_MAXE 4
template <typename T>
__global__ void softmax_window_bias_kernel(T* __restrict__ s, const float* __restrict__ table, const int* __restrict__ rel_index,
                                           const float* __restrict__ mask, long rows, int n, long ld, int heads, int n_mask, float alpha) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int i = (int)(row % n);
  const long wh = row / n;
  const int h = (int)(wh % heads);
  const long w = wh / heads;
  T* sr = s + (size_t)row * ld;
  const int* ri = rel_index + (size_t)i * n;
  const float* mr = mask ? mask + ((size_t)(w % n_mask) * n + i) * n : nullptr;
  float v[SWIN_MAXE];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < SWIN_MAXE; ++k) {
    const int c = lane + k * 64;
    v[k] = -3.0e38f;
    if (c < n) {
      v[k] = ElemTraits<T>::to_f(sr[c]) * alpha + table[(size_t)ri[c] * heads + h];
      if (mr) v[k] += mr[c];
    }
    mx = fmaxf(mx, v[k]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < SWIN_MAXE; ++k) {
    const int c = lane + k * 64;
    v[k] = (c < n) ? __expf(v[k] - mx) : 0.f;
    sum += v[k];
  }
  const float inv = 1.f / wave_sum(sum);
#pragma unroll
  for (int k = 0; k < SWIN_MAXE; ++k) {
    const int c = lane + k * 64;
    if (c < n) sr[c] = ElemTraits<T>::from_f(v[k] * inv);
  }
}

extern "C" int tfpp_softmax_window_bias(void* s, const float* table, const int32_t* rel_index, const float* mask, int64_t windows, int heads,
                                        int n, int64_t ld, int n_mask, float alpha, int dtype, void* stream) {
  if (!s || !table || !rel_index || windows < 1 || heads < 1 || n < 1 || n > 64 * SWIN_MAXE || ld < n || (mask && n_mask < 1)) return TFPP_EINVAL;
  const long rows = (long)windows * heads * n;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(softmax_window_bias_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)s, table, (const int*)rel_index, mask, rows,
                       n, (long)ld, heads, mask ? n_mask : 1, alpha);
  else
    hipLaunchKernelGGL(softmax_window_bias_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (bf16_t*)s, table, (const int*)rel_index, mask,
                       rows, n, (long)ld, heads, mask ? n_mask : 1, alpha);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Stochastic depth (timm DropPath, video_swin_transformer.py:216,276-281): y[b] = x[b] * (keep_b / (1 - p)), keep_b ~ Bernoulli(1 - p) per
// SAMPLE, drawn from the dropout hash on (seed + per-step device counter, b) so that backward (the same call on the gradient) and hipGraph
// replays see the right masks.  rows_per_sample rows of C channels per sample.
template <typename T>
__global__ void drop_path_kernel(const T* __restrict__ x, T* __restrict__ y, long n_chunks, long chunks_per_sample, float p, float inv_keep,
                                 unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_chunks) return;
  const float s = dropout_scale(seed, (unsigned long long)(i / chunks_per_sample), p, inv_keep);
  float v[VEC];
  load_vec<T>(x + i * VEC, v);
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] *= s;
  store_vec<T>(y + i * VEC, v);
}

extern "C" int tfpp_drop_path(const void* x, void* y, int64_t samples, int64_t elems_per_sample, float p, uint64_t seed, const uint64_t* seed_offset,
                              int dtype, void* stream) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (!x || !y || samples < 1 || elems_per_sample < VEC || elems_per_sample % VEC || p < 0.f || p >= 1.f) return TFPP_EINVAL;
  const long n = (long)samples * (elems_per_sample / VEC);
  dim3 grid((unsigned)((n + 255) / 256));
  const float inv_keep = 1.f / (1.f - p);
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(drop_path_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, (long)(elems_per_sample / VEC), p,
                       inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  else
    hipLaunchKernelGGL(drop_path_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, n, (long)(elems_per_sample / VEC), p,
                       inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Gradient of relative_position_bias_table: dtable[rel_index[i][j]][h] += scale * sum_w ds[w][h][i][j] (ds = gradient of the pre-softmax
// scores as tfpp_softmax_bwd leaves it, i.e. already multiplied by alpha: scale = 1 / alpha).  One thread per (h, i, j): the sum over
// windows is a strided walk (consecutive lanes = consecutive j) into a dense (heads, n, n) fp32 image; a second kernel adds, per table entry, the
// pairs that point at it in the order of an inverse index built once per geometry on the host (round 6: the fp32 atomics of rounds 2-5 are gone,
// the gradient is bit-reproducible).
template <typename T>
__global__ void window_bias_dense_kernel(const T* __restrict__ ds, float* __restrict__ dense, long windows, int heads, int n, long ld) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)heads * n * n) return;
  const int j = (int)(t % n);
  const int i = (int)((t / n) % n);
  const int h = (int)(t / ((long)n * n));
  const T* p = ds + ((size_t)h * n + i) * ld + j;
  const size_t wstride = (size_t)heads * n * ld;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four independent chains over the windows (fixed association: bit-reproducible)
  long w = 0;
  for (; w + 3 < windows; w += 4) {
    a0 += ElemTraits<T>::to_f(p[w * wstride]);
    a1 += ElemTraits<T>::to_f(p[(w + 1) * wstride]);
    a2 += ElemTraits<T>::to_f(p[(w + 2) * wstride]);
    a3 += ElemTraits<T>::to_f(p[(w + 3) * wstride]);
  }
  for (; w < windows; ++w) a0 += ElemTraits<T>::to_f(p[w * wstride]);
  dense[t] = (a0 + a1) + (a2 + a3);
}
// dtable[t][h] += scale * sum over the (i, j) pairs of table entry t, in the order of the inverse index (single writer per cell: no atomics)
__global__ void window_bias_gather_kernel(const float* __restrict__ dense, const int* __restrict__ inv_ptr, const int* __restrict__ inv_pairs,
                                          float* __restrict__ dtable, int ntab, int heads, int n, float scale) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= ntab * heads) return;
  const int t = q / heads, h = q - t * heads;
  const float* d = dense + (size_t)h * n * n;
  float acc = 0.f;
  for (int k = inv_ptr[t]; k < inv_ptr[t + 1]; ++k) acc += d[inv_pairs[k]];
  dtable[(size_t)t * heads + h] += acc * scale;
}

extern "C" int tfpp_window_bias_grad(const void* ds, const int32_t* inv_ptr, const int32_t* inv_pairs, int ntab, float* dense_scratch, float* dtable,
                                     int64_t windows, int heads, int n, int64_t ld, float scale, int dtype, void* stream) {
  if (!ds || !inv_ptr || !inv_pairs || !dense_scratch || !dtable || ntab < 1 || windows < 1 || heads < 1 || n < 1 || ld < n) return TFPP_EINVAL;
  const long total = (long)heads * n * n;
  dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(window_bias_dense_kernel<float>, grid, dim3(256), 0, st, (const float*)ds, dense_scratch, (long)windows, heads, n, (long)ld);
  else
    hipLaunchKernelGGL(window_bias_dense_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)ds, dense_scratch, (long)windows, heads, n, (long)ld);
  hipLaunchKernelGGL(window_bias_gather_kernel, dim3((unsigned)((ntab * heads + 255) / 256)), dim3(256), 0, st, dense_scratch, (const int*)inv_ptr,
                     (const int*)inv_pairs, dtable, ntab, heads, n, scale);
  TFPP_CHECK_LAUNCH();
  return 0;
}
