// Normalisation kernels: BatchNorm2d (train statistics / fold / backward), LayerNorm and row softmax (+dropout).
// BatchNorm reductions accumulate per-thread partials in fp32 over <=64 rows and combine in double (native fp64
// atomics on gfx950) so that E[x^2]-E[x]^2 stays accurate; LayerNorm / softmax use one wave64 per row with
// shuffle reductions.
#include "common.cuh"
#include "../../include/tfpp.h"

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm reductions, two stages, no atomics.
// Stage 1: block = 256 threads = CVB channel-vectors x NRS row-slots (CVB = pow2 >= min(CV,64)); a block walks its rows
// with a grid stride (4 independent 16-byte loads in flight per lane) and writes one partial per channel to
// partial[blockIdx.x][2*C].  Stage 2 sums the <=512 partials per value in double.
// MODE 0: sum x, sum x^2.   MODE 1 (backward): g = dy*(y>0?), sum g, sum g*xhat.
// ---------------------------------------------------------------------------------------------------------------
#define BN_MAX_PARTIALS 512
template <typename T, int MODE>
__global__ void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                 const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ partial, long rows,
                                 int C, int cvb_log2, int relu_mask) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC, cvb = 1 << cvb_log2, nrs = 256 >> cvb_log2;
  const int cvl = threadIdx.x & (cvb - 1), rs = threadIdx.x >> cvb_log2;
  const int cv = blockIdx.y * cvb + cvl;
  float s0[VEC], s1[VEC], mu[VEC], is[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; is[e] = 1.f; }
  if (cv < CV) {
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { mu[e] = mean[cv * VEC + e]; is[e] = invstd[cv * VEC + e]; }
    }
    const long stride = (long)gridDim.x * nrs;
    constexpr int U = 4;
    long r = (long)blockIdx.x * nrs + rs;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
      uint4 xv[U], gv[U], ov[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = (size_t)(r + u * stride) * C + cv * VEC;
        xv[u] = *reinterpret_cast<const uint4*>(x + off);
        if (MODE == 1) {
          gv[u] = *reinterpret_cast<const uint4*>(dy + off);
          if (relu_mask) ov[u] = *reinterpret_cast<const uint4*>(y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[VEC];
        unpack16<T>(xv[u], v);
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) { s0[e] += v[e]; s1[e] += v[e] * v[e]; }
        } else {
          float g[VEC];
          unpack16<T>(gv[u], g);
          if (relu_mask) {
            float o[VEC];
            unpack16<T>(ov[u], o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) { s0[e] += g[e]; s1[e] += g[e] * (v[e] - mu[e]) * is[e]; }
        }
      }
    }
    for (; r < rows; r += stride) {
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
    float* out = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < nrs; ++k) { a += sm[k * cvb + cvl][e]; b += sm[k * cvb + cvl][VEC + e]; }
      out[cv * VEC + e] = a;
      out[C + cv * VEC + e] = b;
    }
  }
}

// ws[v] = sum_k partial[k][v] in double; block = 64 values x 4 slots
__global__ void bn_reduce_final_kernel(const float* __restrict__ partial, double* __restrict__ ws, int nblk, int n2c) {
  const int vl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int v = blockIdx.x * 64 + vl;
  double s = 0.0;
  if (v < n2c)
    for (int k = slot; k < nblk; k += 4) s += (double)partial[(size_t)k * n2c + v];
  __shared__ double sm[4][64];
  sm[slot][vl] = s;
  __syncthreads();
  if (slot == 0 && v < n2c) ws[v] = sm[0][vl] + sm[1][vl] + sm[2][vl] + sm[3][vl];
}

template <typename T, int MODE>
static int launch_bn_reduce(const void* x, const void* dy, const void* y, const float* mean, const float* invstd, float* partial, double* ws,
                            long rows, int C, int relu_mask, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  const int CV = C / VEC;
  int lg = 0;
  while ((1 << lg) < CV && lg < 6) ++lg;
  const int cvb = 1 << lg, nrs = 256 >> lg;
  long nblk = (rows + (long)nrs * 8 - 1) / ((long)nrs * 8);  // >= 8 rows per thread
  const int ny = (CV + cvb - 1) / cvb;
  const long cap = BN_MAX_PARTIALS;
  if (nblk > cap) nblk = cap;
  if (nblk * ny > 4096) nblk = 4096 / ny > 1 ? 4096 / ny : 1;
  if (nblk < 1) nblk = 1;
  dim3 grid((unsigned)nblk, (unsigned)ny);
  hipLaunchKernelGGL((bn_reduce_kernel<T, MODE>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, invstd, partial, rows, C,
                     lg, relu_mask);
  hipLaunchKernelGGL(bn_reduce_final_kernel, dim3((2 * C + 63) / 64), dim3(256), 0, st, partial, ws, (int)nblk, 2 * C);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_bn_reduce_final(const float* partial, double* ws, int nblk, int n2c, void* stream) {
  if (!partial || !ws || nblk < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(bn_reduce_final_kernel, dim3((n2c + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, ws, nblk, n2c);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_bn_scratch_floats(int C) { return BN_MAX_PARTIALS * 2 * C + 4 * C; }

extern "C" int tfpp_bn_stats(const void* x, float* scratch, double* ws, int64_t rows, int C, int dtype, void* stream) {
  if (!x || !ws || !scratch) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_reduce<float, 0>(x, nullptr, nullptr, nullptr, nullptr, scratch, ws, (long)rows, C, 0, st)
                           : launch_bn_reduce<bf16_t, 0>(x, nullptr, nullptr, nullptr, nullptr, scratch, ws, (long)rows, C, 0, st);
}

extern "C" int tfpp_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* save_mean, const float* save_invstd,
                                  float* scratch, double* ws, int64_t rows, int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !ws || !scratch || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_reduce<float, 1>(x, dy, y, save_mean, save_invstd, scratch, ws, (long)rows, C, relu_mask, st)
                           : launch_bn_reduce<bf16_t, 1>(x, dy, y, save_mean, save_invstd, scratch, ws, (long)rows, C, relu_mask, st);
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

// dx = gamma*invstd*(g - ws0/rows - xhat*ws1/rows) = A[c]*g + Bc[c]*x + D[c] ; dres = g.
// coefficient kernel (per channel) also accumulates dgamma += ws1, dbeta += ws0.
__global__ void bn_bwd_coef_kernel(const double* __restrict__ ws, const float* __restrict__ gamma, const float* __restrict__ mean,
                                   const float* __restrict__ invstd, float* __restrict__ coef, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, long rows, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double n = (double)rows, s0 = ws[c], s1 = ws[C + c];
  const double gm = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c], mu = (double)mean[c];
  const double A = gm * is, Bc = -gm * is * is * s1 / n, D = -gm * is * s0 / n - Bc * mu;
  coef[c] = (float)A;
  coef[C + c] = (float)Bc;
  coef[2 * C + c] = (float)D;
  if (dgamma) dgamma[c] += (float)s1;
  if (dbeta) dbeta[c] += (float)s0;
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ coef,
                                    T* __restrict__ dx, T* __restrict__ dres, long nvec, int C, int relu_mask) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
    for (int e = 0; e < VEC; ++e) v[e] = coef[c0 + e] * g[e] + coef[C + c0 + e] * v[e] + coef[2 * C + c0 + e];
    store_vec<T>(dx + i * VEC, v);
  }
}

extern "C" int tfpp_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                                 const float* save_invstd, const double* ws, float* scratch, void* dx, void* dres, float* dgamma,
                                 float* dbeta, int64_t rows, int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !ws || !dx || !scratch || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC) return TFPP_EINVAL;
  float* coef = scratch + (size_t)BN_MAX_PARTIALS * 2 * C;  // after the stage-1 partials
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, ws, gamma, save_mean, save_invstd, coef, dgamma, dbeta,
                     (long)rows, C);
  long nvec = rows * (C / VEC);
  long blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)dy, (const float*)y, (const float*)x, coef, (float*)dx, (float*)dres, nvec, C, relu_mask);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (const bf16_t*)x, coef, (bf16_t*)dx, (bf16_t*)dres, nvec, C, relu_mask);
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

// parameter gradients of LayerNorm as a column reduction: dgamma[c] += sum_r dy*xhat, dbeta[c] += sum_r dy.
// grid (ceil(C/64), S); block = 64 channels x 4 row slots; <= S atomics per address.
template <typename T>
__global__ void layernorm_param_grad_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                            long rows, int C, long rows_per_block) {
  const int cl = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float sg = 0.f, sb = 0.f;
  if (c < C) {
    for (long r = r0 + slot; r < r1; r += 4) {
      const float g = ElemTraits<T>::to_f(dy[(size_t)r * C + c]);
      const float xh = (ElemTraits<T>::to_f(x[(size_t)r * C + c]) - mean[r]) * rstd[r];
      sg += g * xh;
      sb += g;
    }
  }
  __shared__ float sm[2][4][64];
  sm[0][slot][cl] = sg;
  sm[1][slot][cl] = sb;
  __syncthreads();
  if (slot == 0 && c < C) {
    if (dgamma) atomicAdd(dgamma + c, sm[0][0][cl] + sm[0][1][cl] + sm[0][2][cl] + sm[0][3][cl]);
    if (dbeta) atomicAdd(dbeta + c, sm[1][0][cl] + sm[1][1][cl] + sm[1][2][cl] + sm[1][3][cl]);
  }
}

extern "C" int tfpp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                  float* dgamma, float* dbeta, int64_t rows, int C, int dtype, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC || C / VEC > 64 * 6) return TFPP_EINVAL;
  const int nv = (C / VEC + 63) / 64;
  const int rpw = 1;  // dx: one wave per row, no atomics (parameter gradients come from the column-reduction kernel below)
  const long waves = (rows + rpw - 1) / rpw;
  dim3 grid((unsigned)((waves + 3) / 4));
  float* nullf = nullptr;
#define LN_BWD(TT, MV) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, MV>), grid, dim3(256), 0, st, (const TT*)dy, (const TT*)x, gamma, mean, rstd, (TT*)dx, nullf, nullf, (long)rows, C, rpw)
#define LN_BWD_T(TT) do { if (nv <= 1) LN_BWD(TT, 1); else if (nv <= 2) LN_BWD(TT, 2); else if (nv <= 3) LN_BWD(TT, 3); else if (nv <= 4) LN_BWD(TT, 4); else LN_BWD(TT, 6); } while (0)
  if (dtype == TFPP_F32) LN_BWD_T(float); else LN_BWD_T(bf16_t);
  if (dgamma || dbeta) {
    long S = rows / 16;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    const long rpb = (rows + S - 1) / S;
    dim3 g2((unsigned)((C + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
    if (dtype == TFPP_F32) hipLaunchKernelGGL(layernorm_param_grad_kernel<float>, g2, dim3(256), 0, st, (const float*)dy, (const float*)x, mean, rstd, dgamma, dbeta, (long)rows, C, rpb);
    else hipLaunchKernelGGL(layernorm_param_grad_kernel<bf16_t>, g2, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, dgamma, dbeta, (long)rows, C, rpb);
  }
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
                                   unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;  // per-step device counter: hipGraph replays draw fresh masks
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

extern "C" int tfpp_softmax_fwd(void* x, void* pd, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype,
                                void* stream) {
  if (!x || cols > 64 * SM_MAXE) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(softmax_fwd_kernel<float>, grid, dim3(256), 0, st, (float*)x, (float*)pd, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  else hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (bf16_t*)x, (bf16_t*)pd, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// in place on dp: dP = dPd * mask ; dS = alpha * P .* (dP - sum_j dP_j P_j)
template <typename T>
__global__ void softmax_bwd_kernel(const T* __restrict__ p, T* __restrict__ dp, long rows, int cols, long ld, float alpha, float p_drop,
                                   float inv_keep, unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;  // per-step device counter: hipGraph replays draw fresh masks
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

extern "C" int tfpp_softmax_bwd(const void* p, void* dp_inout, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                                int dtype, void* stream) {
  if (!p || !dp_inout || cols > 64 * SM_MAXE) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(softmax_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)p, (float*)dp_inout, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  else hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)p, (bf16_t*)dp_inout, (long)rows, cols, (long)ld, alpha, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}
