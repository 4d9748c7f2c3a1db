// Normalisation kernels: BatchNorm2d (train statistics / fold / backward), LayerNorm and row softmax (+dropout).
// BatchNorm reductions accumulate per-thread partials in fp32 over <=64 rows and combine in double (native fp64
// atomics on gfx950) so that E[x^2]-E[x]^2 stays accurate; LayerNorm / softmax use one wave64 per row with
// shuffle reductions.
#include "common.cuh"
#include "../../include/tfpp.h"

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm statistics.  Block = 256 threads = CVB channel-vectors x RS row-slots (CVB = pow2 >= min(CV,64)).
// MODE 0: sum x, sum x^2.   MODE 1 (backward): g = dy*(y>0?), sum g, sum g*xhat.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                 const float* __restrict__ mean, const float* __restrict__ invstd, double* __restrict__ ws, long rows,
                                 int C, int cvb_log2, int relu_mask, int rows_per_block) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC, cvb = 1 << cvb_log2, nrs = 256 >> cvb_log2;
  const int cvl = threadIdx.x & (cvb - 1), rs = threadIdx.x >> cvb_log2;
  const int cv = blockIdx.y * cvb + cvl;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s0[VEC], s1[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
  if (cv < CV) {
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { mu[e] = mean[cv * VEC + e]; is[e] = invstd[cv * VEC + e]; }
    }
    for (long r = r0 + rs; r < r1; r += nrs) {
      const size_t off = (size_t)r * C + cv * VEC;
      float v[VEC];
      load_vec<T>(x + off, v);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s0[e] += v[e]; s1[e] += v[e] * v[e]; }
      } else {
        float g[VEC];
        load_vec<T>(dy + off, g);
        if (relu_mask) {
          float o[VEC];
          load_vec<T>(y + off, o);
#pragma unroll
          for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { s0[e] += g[e]; s1[e] += g[e] * (v[e] - mu[e]) * is[e]; }
      }
    }
  }
  __shared__ float sm[256][2 * VEC + 1];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sm[threadIdx.x][e] = s0[e]; sm[threadIdx.x][VEC + e] = s1[e]; }
  __syncthreads();
  if (rs == 0 && cv < CV) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < nrs; ++k) { a += (double)sm[k * cvb + cvl][e]; b += (double)sm[k * cvb + cvl][VEC + e]; }
      atomicAdd(ws + cv * VEC + e, a);
      atomicAdd(ws + C + cv * VEC + e, b);
    }
  }
}

template <typename T, int MODE>
static int launch_bn_reduce(const void* x, const void* dy, const void* y, const float* mean, const float* invstd, double* ws, long rows, int C,
                            int relu_mask, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)2 * C * sizeof(double), st);
  if (e != hipSuccess) return -(int)e;
  const int CV = C / VEC;
  int lg = 0;
  while ((1 << lg) < CV && lg < 6) ++lg;
  const int cvb = 1 << lg, nrs = 256 >> lg;
  const int rpb = nrs * 16;  // 16 rows per thread
  dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((CV + cvb - 1) / cvb));
  hipLaunchKernelGGL((bn_reduce_kernel<T, MODE>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, invstd, ws, rows, C, lg,
                     relu_mask, rpb);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_bn_stats(const void* x, double* ws, int64_t rows, int C, int dtype, void* stream) {
  if (!x || !ws) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_reduce<float, 0>(x, nullptr, nullptr, nullptr, nullptr, ws, (long)rows, C, 0, st)
                           : launch_bn_reduce<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, ws, (long)rows, C, 0, st);
}

extern "C" int tfpp_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* save_mean, const float* save_invstd, double* ws,
                                  int64_t rows, int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !ws || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_reduce<float, 1>(x, dy, y, save_mean, save_invstd, ws, (long)rows, C, relu_mask, st)
                           : launch_bn_reduce<bf16_t, 1>(x, dy, y, save_mean, save_invstd, ws, (long)rows, C, relu_mask, st);
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rm, float* __restrict__ rv, long long* __restrict__ nbt, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ save_mean, float* __restrict__ save_invstd, long rows, int C,
                                   float momentum, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double n = (double)rows;
  const double m = ws[c] / n;
  double var = ws[C + c] / n - m * m;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - (float)m * g * invstd;
  if (save_mean) save_mean[c] = (float)m;
  if (save_invstd) save_invstd[c] = invstd;
  if (rm) rm[c] = (1.f - momentum) * rm[c] + momentum * (float)m;
  if (rv) {
    const double unb = rows > 1 ? var * n / (n - 1.0) : var;
    rv[c] = (1.f - momentum) * rv[c] + momentum * (float)unb;
  }
}

extern "C" int tfpp_bn_finalize(const double* ws, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean, float* save_invstd, int64_t rows,
                                int C, float momentum, float eps, void* stream) {
  if (!ws || !scale || !shift) return TFPP_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, gamma, beta, running_mean, running_var,
                     (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd, (long)rows, C, momentum, eps);
  TFPP_CHECK_LAUNCH();
  return 0;
}

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                               const float* __restrict__ rv, float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * invstd;
  shift[c] = b - rm[c] * g * invstd;
}

extern "C" int tfpp_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* scale,
                            float* shift, int C, float eps, void* stream) {
  if (!running_mean || !running_var || !scale || !shift) return TFPP_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, scale,
                     shift, C, eps);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dx = gamma*invstd*(g - ws0/rows - xhat*ws1/rows) ; dres = g ; block (0,*) also accumulates dgamma/dbeta
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ gamma,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const double* __restrict__ ws,
                                    T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta, long rows,
                                    int C, int relu_mask) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long nvec = rows * CV;
  const float invn = 1.f / (float)rows;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C) {
    if (dgamma) dgamma[i] += (float)ws[C + i];
    if (dbeta) dbeta[i] += (float)ws[i];
  }
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < nvec; i += stride) {
    const int c0 = (int)(i % CV) * VEC;
    float g[VEC], v[VEC];
    load_vec<T>(dy + i * VEC, g);
    load_vec<T>(x + i * VEC, v);
    if (relu_mask) {
      float o[VEC];
      load_vec<T>(y + i * VEC, o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (dres) store_vec<T>(dres + i * VEC, g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = c0 + e;
      const float is = invstd[c], xh = (v[e] - mean[c]) * is;
      const float gm = gamma ? gamma[c] : 1.f;
      v[e] = gm * is * (g[e] - (float)ws[c] * invn - xh * (float)ws[C + c] * invn);
    }
    store_vec<T>(dx + i * VEC, v);
  }
}

extern "C" int tfpp_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                                 const float* save_invstd, const double* ws, void* dx, void* dres, float* dgamma, float* dbeta, int64_t rows,
                                 int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !ws || !dx || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC) return TFPP_EINVAL;
  long nvec = rows * (C / VEC);
  long blocks = (nvec + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks * 256 < C) blocks = (C + 255) / 256;
  // NOTE: the dgamma/dbeta accumulation uses the first C global threads; grid-stride keeps them valid.
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)x, gamma, save_mean, save_invstd, ws, (float*)dx, (float*)dres, dgamma, dbeta, (long)rows, C, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, gamma, save_mean, save_invstd, ws, (bf16_t*)dx, (bf16_t*)dres, dgamma, dbeta, (long)rows, C, relu_mask);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row kept in registers (C <= 64 * MAXV * VEC)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int LN_MAXV>
__global__ void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ y,
                                     float* __restrict__ mean_o, float* __restrict__ rstd_o, long rows, int C, float eps) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[LN_MAXV][VEC];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
      load_vec<T>(x + (size_t)row * C + cv * VEC, v[k]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += v[k][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = v[k][e] - mean; q += d * d; }
    }
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int cv = lane + k * 64;
    if (cv < CV) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = (v[k][e] - mean) * rstd * gamma[cv * VEC + e] + beta[cv * VEC + e];
      store_vec<T>(y + (size_t)row * C + cv * VEC, o);
    }
  }
}

extern "C" int tfpp_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int C,
                                  float eps, int dtype, void* stream) {
  if (!x || !y || !gamma || !beta) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC || C / VEC > 64 * 6) return TFPP_EINVAL;
  const int nv = (C / VEC + 63) / 64;
  dim3 grid((unsigned)((rows + 3) / 4));
#define LN_FWD(TT, MV) hipLaunchKernelGGL((layernorm_fwd_kernel<TT, MV>), grid, dim3(256), 0, st, (const TT*)x, gamma, beta, (TT*)y, mean, rstd, (long)rows, C, eps)
#define LN_FWD_T(TT) do { if (nv <= 1) LN_FWD(TT, 1); else if (nv <= 2) LN_FWD(TT, 2); else if (nv <= 3) LN_FWD(TT, 3); else if (nv <= 4) LN_FWD(TT, 4); else LN_FWD(TT, 6); } while (0)
  if (dtype == TFPP_F32) LN_FWD_T(float); else LN_FWD_T(bf16_t);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// each wave walks rows_per_wave rows, accumulating dgamma/dbeta partials in registers, one atomic per channel at the end
template <typename T, int LN_MAXV>
__global__ void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C, int rows_per_wave) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long r0 = w * rows_per_wave;
  float ag[LN_MAXV][VEC], ab[LN_MAXV][VEC], gm[LN_MAXV][VEC];
#pragma unroll
  for (int k = 0; k < LN_MAXV; ++k) {
    const int cv = lane + k * 64;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; gm[k][e] = (cv < CV) ? gamma[cv * VEC + e] : 0.f; }
  }
  for (long row = r0; row < r0 + rows_per_wave && row < rows; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXV][VEC], xh[LN_MAXV][VEC];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        float v[VEC];
        load_vec<T>(dy + (size_t)row * C + cv * VEC, g[k]);
        load_vec<T>(x + (size_t)row * C + cv * VEC, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[k][e] = (v[e] - mu) * rs;
          ag[k][e] += g[k][e] * xh[k][e];
          ab[k][e] += g[k][e];
          g[k][e] *= gm[k][e];
          c1 += g[k][e];
          c2 += g[k][e] * xh[k][e];
        }
      }
    }
    c1 = wave_sum(c1) / (float)C;
    c2 = wave_sum(c2) / (float)C;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rs * (g[k][e] - c1 - xh[k][e] * c2);
        store_vec<T>(dx + (size_t)row * C + cv * VEC, o);
      }
    }
  }
  if (r0 < rows) {
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (dgamma) atomicAdd(dgamma + cv * VEC + e, ag[k][e]);
          if (dbeta) atomicAdd(dbeta + cv * VEC + e, ab[k][e]);
        }
      }
    }
  }
}

extern "C" int tfpp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                  float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC || C / VEC > 64 * 6) return TFPP_EINVAL;
  const int nv = (C / VEC + 63) / 64;
  const int rpw = rows >= 2048 ? 8 : 1;
  const long waves = (rows + rpw - 1) / rpw;
  dim3 grid((unsigned)((waves + 3) / 4));
#define LN_BWD(TT, MV) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, MV>), grid, dim3(256), 0, st, (const TT*)dy, (const TT*)x, gamma, mean, rstd, (TT*)dx, dgamma, dbeta, (long)rows, C, rpw)
#define LN_BWD_T(TT) do { if (nv <= 1) LN_BWD(TT, 1); else if (nv <= 2) LN_BWD(TT, 2); else if (nv <= 3) LN_BWD(TT, 3); else if (nv <= 4) LN_BWD(TT, 4); else LN_BWD(TT, 6); } while (0)
  if (dtype == TFPP_F32) LN_BWD_T(float); else LN_BWD_T(bf16_t);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// row softmax (one wave per row, cols <= 64*SM_MAXE), optional dropout copy
// ---------------------------------------------------------------------------------------------------------------
#define SM_MAXE 8
// P = softmax(alpha * x) written in place; if pd != NULL also pd = dropout(P) (mask keyed by seed and element index)
template <typename T>
__global__ void softmax_fwd_kernel(T* __restrict__ x, T* __restrict__ pd, long rows, int cols, long ld, float alpha, float p_drop, float inv_keep,
                                   unsigned long long seed) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* xr = x + (size_t)row * ld;
  float v[SM_MAXE];
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < SM_MAXE; ++k) {
    const int c = lane + k * 64;
    v[k] = (c < cols) ? ElemTraits<T>::to_f(xr[c]) * alpha : -3.0e38f;
    mx = fmaxf(mx, v[k]);
  }
  mx = wave_max(mx);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXE; ++k) {
    const int c = lane + k * 64;
    v[k] = (c < cols) ? __expf(v[k] - mx) : 0.f;
    s += v[k];
  }
  const float inv = 1.f / wave_sum(s);
#pragma unroll
  for (int k = 0; k < SM_MAXE; ++k) {
    const int c = lane + k * 64;
    if (c < cols) {
      const float pv = v[k] * inv;
      xr[c] = ElemTraits<T>::from_f(pv);
      if (pd) pd[(size_t)row * ld + c] = ElemTraits<T>::from_f(pv * dropout_scale(seed, (unsigned long long)row * cols + c, p_drop, inv_keep));
    }
  }
}

extern "C" int tfpp_softmax_fwd(void* x, void* pd, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed, int dtype,
                                void* stream) {
  if (!x || cols > 64 * SM_MAXE) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, grid, dim3(256), 0, st, (float*)x, (float*)pd, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed);
  else hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (bf16_t*)x, (bf16_t*)pd, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// in place on dp: dP = dPd * mask ; dS = alpha * P .* (dP - sum_j dP_j P_j)
template <typename T>
__global__ void softmax_bwd_kernel(const T* __restrict__ p, T* __restrict__ dp, long rows, int cols, long ld, float alpha, float p_drop,
                                   float inv_keep, unsigned long long seed) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* pr = p + (size_t)row * ld;
  T* dr = dp + (size_t)row * ld;
  float pv[SM_MAXE], g[SM_MAXE];
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < SM_MAXE; ++k) {
    const int c = lane + k * 64;
    pv[k] = 0.f; g[k] = 0.f;
    if (c < cols) {
      pv[k] = ElemTraits<T>::to_f(pr[c]);
      g[k] = ElemTraits<T>::to_f(dr[c]);
      if (p_drop > 0.f) g[k] *= dropout_scale(seed, (unsigned long long)row * cols + c, p_drop, inv_keep);
      dot += g[k] * pv[k];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int k = 0; k < SM_MAXE; ++k) {
    const int c = lane + k * 64;
    if (c < cols) dr[c] = ElemTraits<T>::from_f(alpha * pv[k] * (g[k] - dot));
  }
}

extern "C" int tfpp_softmax_bwd(const void* p, void* dp_inout, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed,
                                int dtype, void* stream) {
  if (!p || !dp_inout || cols > 64 * SM_MAXE) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(softmax_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)p, (float*)dp_inout, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed);
  else hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)p, (bf16_t*)dp_inout, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed);
  TFPP_CHECK_LAUNCH();
  return 0;
}
