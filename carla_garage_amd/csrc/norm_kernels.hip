// Normalisation kernels: BatchNorm2d (train statistics / fold / backward), LayerNorm and row softmax (+dropout).
// BatchNorm reductions accumulate per-thread partials in fp32 over <=64 rows and combine in double (native fp64
// atomics on gfx950) so that E[x^2]-E[x]^2 stays accurate; LayerNorm / softmax use one wave64 per row with
// shuffle reductions.
#include "common.h"
#include "../../include/tfpp.h"

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm reductions, two stages, no atomics.
// Stage 1: column-fixed layout (common.h): a workgroup is RP row-slots x SW channel-chunks and walks its rows with a
// grid stride (U independent 16-byte loads per operand in flight per lane), then writes one partial per channel to
// partial[blockIdx.x][2*C].  Stage 2 sums the <=512 partials per value in double.
// MODE 0: sum x, sum x^2.   MODE 1 (backward): g = dy*(y>0?), sum g, sum g*xhat.
// ---------------------------------------------------------------------------------------------------------------
#define BN_MAX_PARTIALS 512
template <typename T, int MODE>
__global__ void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                 const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ partial, long rows,
                                 int C, int sw, int rp, int relu_mask) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  int rr, cv;
  const bool active = col_thread(sw, rp, CV, rr, cv);
  float acc[2 * VEC];  // [0,VEC): s0   [VEC,2VEC): s1
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) acc[e] = 0.f;
  if (active) {
    float mu[VEC], is[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mu[e] = MODE == 1 ? mean[cv * VEC + e] : 0.f; is[e] = MODE == 1 ? invstd[cv * VEC + e] : 1.f; }
    auto accumulate = [&](const uint4& xv, const uint4& gv, const uint4& ov) {
      float v[VEC];
      unpack16<T>(xv, v);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc[e] += v[e]; acc[VEC + e] += v[e] * v[e]; }
      } else {
        float g[VEC];
        unpack16<T>(gv, g);
        if (relu_mask) {
          float o[VEC];
          unpack16<T>(ov, o);
#pragma unroll
          for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc[e] += g[e]; acc[VEC + e] += g[e] * (v[e] - mu[e]) * is[e]; }
      }
    };
    const long stride = (long)gridDim.x * rp;
    constexpr int U = MODE == 0 ? 4 : 2;
    long r = (long)blockIdx.x * rp + rr;
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
      for (int u = 0; u < U; ++u) accumulate(xv[u], gv[u], ov[u]);
    }
    for (; r < rows; r += stride) {
      const size_t off = (size_t)r * C + cv * VEC;
      uint4 xv = *reinterpret_cast<const uint4*>(x + off), gv = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0);
      if (MODE == 1) {
        gv = *reinterpret_cast<const uint4*>(dy + off);
        if (relu_mask) ov = *reinterpret_cast<const uint4*>(y + off);
      }
      accumulate(xv, gv, ov);
    }
  }
  __shared__ float sm[2 * VEC * 256];
  col_block_reduce<2 * VEC>(acc, sw, rp, rr, sm);
  if (active && rr == 0) {
    float* out = partial + (size_t)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      out[cv * VEC + e] = acc[e];
      out[C + cv * VEC + e] = acc[VEC + e];
    }
  }
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Column sums of the partial rows (sum half | sum-of-products half, row-major [nrows][2C]) in double, COALESCED and in a fixed order: a
// workgroup of NT threads owns 32 channels; thread (rg, cl) walks rows rg, rg + RG, ... (RG = NT / 32) reading the 128-byte segments
// [c0, c0 + 32) of both halves, so a wave touches 4 cache lines per row pair instead of the 64 lines per load of a lane-per-row walk of one
// column (round 2: the lane-per-row finalize cost 9.4 us per BatchNorm layer, 136 launches on the critical chain).  The RG partial sums are
// combined through LDS by the rg = 0 threads in index order: independent of scheduling, bit-reproducible.  clear: re-zero what was read (the
// conv epilogue accumulates into rows that are zero between uses, so no memset launch is needed).  Returns true in the threads that hold a
// channel's totals.
template <int NT>
__device__ __forceinline__ bool partial_cols_sum(float* __restrict__ partial, int nrows, int C, bool clear, int& c, double& s0, double& s1) {
  constexpr int RG = NT / 32;
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  c = (int)blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int k = rg;
    for (; k + 3 * RG < nrows; k += 4 * RG) {
      float va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float* row = partial + (size_t)(k + u * RG) * 2 * C;
        va[u] = row[c];
        vb[u] = row[C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) { a += (double)va[u]; b += (double)vb[u]; }
      if (clear) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float* row = partial + (size_t)(k + u * RG) * 2 * C;
          row[c] = 0.f;
          row[C + c] = 0.f;
        }
      }
    }
    for (; k < nrows; k += RG) {
      float* row = partial + (size_t)k * 2 * C;
      a += (double)row[c];
      b += (double)row[C + c];
      if (clear) { row[c] = 0.f; row[C + c] = 0.f; }
    }
  }
  __shared__ double sm[2][RG][33];
  sm[0][rg][cl] = a;
  sm[1][rg][cl] = b;
  __syncthreads();
  if (rg != 0 || c >= C) return false;
  s0 = 0.0;
  s1 = 0.0;
#pragma unroll 8
  for (int r = 0; r < RG; ++r) { s0 += sm[0][r][cl]; s1 += sm[1][r][cl]; }
  return true;
}

// ws[v] = sum_k partial[k][v] in double; one wave per value
__global__ void bn_reduce_final_kernel(const float* __restrict__ partial, double* __restrict__ ws, int nblk, int n2c) {
  const int lane = threadIdx.x & 63, v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n2c) return;
  double s = 0.0;
  for (int k = lane; k < nblk; k += 64) s += (double)partial[(size_t)k * n2c + v];
  s = wave_sum_f64(s);
  if (lane == 0) ws[v] = s;
}

static inline int bn_reduce_blocks(long rows, int CV, int mode) {
  const ColLayout l = col_layout(CV);
  int nblk = col_blocks_x(rows, l, mode == 0 ? 8 : 4, 4096);
  return nblk > BN_MAX_PARTIALS ? BN_MAX_PARTIALS : nblk;
}

template <typename T, int MODE>
static int launch_bn_reduce(const void* x, const void* dy, const void* y, const float* mean, const float* invstd, float* partial, double* ws,
                            long rows, int C, int relu_mask, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  const ColLayout l = col_layout(C / VEC);
  const int nblk = bn_reduce_blocks(rows, C / VEC, MODE);
  dim3 grid((unsigned)nblk, (unsigned)l.ny);
  hipLaunchKernelGGL((bn_reduce_kernel<T, MODE>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, (const T*)y, mean, invstd, partial, rows, C,
                     l.sw, l.rp, relu_mask);
  if (ws) hipLaunchKernelGGL(bn_reduce_final_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, st, partial, ws, nblk, 2 * C);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_bn_reduce_final(const float* partial, double* ws, int nblk, int n2c, void* stream) {
  if (!partial || !ws || nblk < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(bn_reduce_final_kernel, dim3((n2c + 3) / 4), dim3(256), 0, (hipStream_t)stream, partial, ws, nblk, n2c);
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
  if (!dy || !x || !scratch || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
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

// The same from accumulation rows (tfpp_conv_params.stats_partial); re-zeroes the rows it consumed.  WIDE = false: one wave
// per channel (<= 256 rows); WIDE = true: one 256-thread workgroup per channel (one row per M-tile on the large feature
// maps: up to 1536 rows), combined through LDS in a fixed order -- the sum is independent of scheduling either way.
template <int NT>
__global__ __launch_bounds__(NT) void bn_finalize_partials_kernel(float* __restrict__ partial, int nrows, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float* __restrict__ rm, float* __restrict__ rv,
                                            long long* __restrict__ nbt, float* __restrict__ scale, float* __restrict__ shift,
                                            float* __restrict__ save_mean, float* __restrict__ save_invstd, long rows, int C, float momentum,
                                            float eps, int clear) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  int c;
  double s0, s1;
  if (!partial_cols_sum<NT>(partial, nrows, C, clear != 0, c, s0, s1)) return;
  const double n = (double)rows;
  const double m = s0 / n;
  double var = s1 / n - m * m;
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

extern "C" int tfpp_bn_finalize_partials(float* partial, int nrows, int clear, const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                                         float* save_invstd, int64_t rows, int C, float momentum, float eps, void* stream) {
  if (!partial || nrows < 1 || !scale || !shift) return TFPP_EINVAL;
  if (nrows > 128)
    hipLaunchKernelGGL(bn_finalize_partials_kernel<1024>, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial, nrows, gamma, beta,
                       running_mean, running_var, (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd, (long)rows, C,
                       momentum, eps, clear);
  else
    hipLaunchKernelGGL(bn_finalize_partials_kernel<256>, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, nrows, gamma, beta,
                       running_mean, running_var, (long long*)num_batches_tracked, scale, shift, save_mean, save_invstd, (long)rows, C,
                       momentum, eps, clear);
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
// WIDE = false: one wave per channel (few rows); WIDE = true: one 256-thread workgroup per channel (one row per M-tile of the kernel
// that produced the gradient: up to a few thousand rows), combined through LDS in a fixed order.
template <int NT>
__global__ __launch_bounds__(NT) void bn_bwd_coef_kernel(float* __restrict__ partial, int nrows, double* __restrict__ ws, const float* __restrict__ gamma,
                                   const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ coef,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C) {
  int c;
  double s0, s1;
  if (!partial_cols_sum<NT>(partial, nrows, C, false, c, s0, s1)) return;
  const double n = (double)rows;
  const double gm = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c], mu = (double)mean[c];
  const double A = gm * is, Bc = -gm * is * is * s1 / n, D = -gm * is * s0 / n - Bc * mu;
  coef[c] = (float)A;
  coef[C + c] = (float)Bc;
  coef[2 * C + c] = (float)D;
  if (ws) { ws[c] = s0; ws[C + c] = s1; }
  if (dgamma) dgamma[c] += (float)s1;
  if (dbeta) dbeta[c] += (float)s0;
}

template <typename T, bool RELU, bool DRES>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ coef,
                                    T* __restrict__ dx, T* __restrict__ dres, long rows, int C, int sw, int rp) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  int rr, cv;
  if (!col_thread(sw, rp, CV, rr, cv)) return;
  const int c0 = cv * VEC;
  float ka[VEC], kb[VEC], kd[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { ka[e] = coef[c0 + e]; kb[e] = coef[C + c0 + e]; kd[e] = coef[2 * C + c0 + e]; }
  auto body = [&](size_t off, const uint4& gv, const uint4& xv, const uint4& ov) {
    float g[VEC], v[VEC];
    unpack16<T>(gv, g);
    unpack16<T>(xv, v);
    if (RELU) {
      float o[VEC];
      unpack16<T>(ov, o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    if (DRES) store_vec<T>(dres + off, g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = ka[e] * g[e] + kb[e] * v[e] + kd[e];
    store_vec<T>(dx + off, v);
  };
  const long stride = (long)gridDim.x * rp;
  constexpr int U = 2;
  long r = (long)blockIdx.x * rp + rr;
  for (; r + (U - 1) * stride < rows; r += U * stride) {
    uint4 gv[U], xv[U], ov[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * stride) * C + c0;
      gv[u] = *reinterpret_cast<const uint4*>(dy + off);
      xv[u] = *reinterpret_cast<const uint4*>(x + off);
      if (RELU) ov[u] = *reinterpret_cast<const uint4*>(y + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) body((size_t)(r + u * stride) * C + c0, gv[u], xv[u], ov[u]);
  }
  for (; r < rows; r += stride) {
    const size_t off = (size_t)r * C + c0;
    uint4 gv = *reinterpret_cast<const uint4*>(dy + off), xv = *reinterpret_cast<const uint4*>(x + off), ov = make_uint4(0, 0, 0, 0);
    if (RELU) ov = *reinterpret_cast<const uint4*>(y + off);
    body(off, gv, xv, ov);
  }
}

template <typename T>
static void launch_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* coef, void* dx, void* dres, long rows, int C,
                                int relu_mask, hipStream_t st) {
  const ColLayout l = col_layout(C / ElemTraits<T>::VEC);
  dim3 grid((unsigned)col_blocks_x(rows, l, 4, 1 << 20), (unsigned)l.ny);
#define BA(R_, D_) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, R_, D_>), grid, dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, coef, (T*)dx, (T*)dres, rows, C, l.sw, l.rp)
  if (relu_mask) { if (dres) BA(true, true); else BA(true, false); }
  else { if (dres) BA(false, true); else BA(false, false); }
#undef BA
}

static void launch_bn_bwd_coef(float* partial, int nrows, double* ws, const float* gamma, const float* mean, const float* invstd, float* coef,
                               float* dgamma, float* dbeta, long rows, int C, hipStream_t st) {
  if (nrows > 128)
    hipLaunchKernelGGL(bn_bwd_coef_kernel<1024>, dim3((C + 31) / 32), dim3(1024), 0, st, partial, nrows, ws, gamma, mean, invstd, coef, dgamma, dbeta,
                       rows, C);
  else
    hipLaunchKernelGGL(bn_bwd_coef_kernel<256>, dim3((C + 31) / 32), dim3(256), 0, st, partial, nrows, ws, gamma, mean, invstd, coef, dgamma, dbeta,
                       rows, C);
}

extern "C" int tfpp_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                                 const float* save_invstd, double* ws, float* scratch, void* dx, void* dres, float* dgamma,
                                 float* dbeta, int64_t rows, int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !dx || !scratch || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC) return TFPP_EINVAL;
  float* coef = scratch + (size_t)BN_MAX_PARTIALS * 2 * C;  // after the stage-1 partials
  launch_bn_bwd_coef(scratch, bn_reduce_blocks((long)rows, C / VEC, 1), ws, gamma, save_mean, save_invstd, coef, dgamma, dbeta, (long)rows, C, st);
  if (dtype == TFPP_F32) launch_bn_bwd_apply<float>(dy, y, x, coef, dx, dres, (long)rows, C, relu_mask, st);
  else launch_bn_bwd_apply<bf16_t>(dy, y, x, coef, dx, dres, (long)rows, C, relu_mask, st);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_bn_bwd_apply_rows(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                                      const float* save_invstd, float* partial, int nrows, float* coef, void* dx, void* dres,
                                      float* dgamma, float* dbeta, int64_t rows, int C, int relu_mask, int dtype, void* stream) {
  if (!dy || !x || !dx || !partial || nrows < 1 || !coef || !save_mean || !save_invstd || (relu_mask && !y)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC) return TFPP_EINVAL;
  launch_bn_bwd_coef(partial, nrows, nullptr, gamma, save_mean, save_invstd, coef, dgamma, dbeta, (long)rows, C, st);
  if (dtype == TFPP_F32) launch_bn_bwd_apply<float>(dy, y, x, coef, dx, dres, (long)rows, C, relu_mask, st);
  else launch_bn_bwd_apply<bf16_t>(dy, y, x, coef, dx, dres, (long)rows, C, relu_mask, st);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// squeeze-excite backward apply with the BatchNorm-backward statistics of the layer in front of it fused in (conv2 of a RegNet
// bottleneck: dx is the complete gradient of a2 = relu(BN2(raw2))):  dx = dy * gate[b,c] + dpool[b,c] / HW, and per workgroup
// one row of  sum g, sum g * xhat  with g = dx (rounded) * (a2 > 0).  Column-fixed layout, one grid z-slice per sample so that
// the gate stays in registers; rows of partial: b * gridDim.x + blockIdx.x.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void se_bwd_apply_bns_kernel(const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool,
                                        const T* __restrict__ y, const T* __restrict__ x, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, T* __restrict__ dx, float* __restrict__ partial, long HW, int C, int sw,
                                        int rp) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  int rr, cv;
  const bool active = col_thread(sw, rp, CV, rr, cv);
  const int b = blockIdx.z;
  float acc[2 * VEC];
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) acc[e] = 0.f;
  if (active) {
    const int c0 = cv * VEC;
    float gt[VEC], dp[VEC], mu[VEC], is[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gt[e] = gate[(size_t)b * C + c0 + e];
      dp[e] = dpool[(size_t)b * C + c0 + e] * inv;
      mu[e] = mean[c0 + e];
      is[e] = invstd[c0 + e];
    }
    const long stride = (long)gridDim.x * rp;
    for (long r = (long)blockIdx.x * rp + rr; r < HW; r += stride) {
      const size_t off = ((size_t)b * HW + r) * C + c0;
      float v[VEC], yv[VEC], xv[VEC];
      load_vec<T>(dy + off, v);
      load_vec<T>(y + off, yv);
      load_vec<T>(x + off, xv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = v[e] * gt[e] + dp[e];
      const uint4 packed = pack16<T>(v);
      *reinterpret_cast<uint4*>(dx + off) = packed;
      unpack16<T>(packed, v);  // the rounded values the BatchNorm backward reads back
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float g = yv[e] > 0.f ? v[e] : 0.f;
        acc[e] += g;
        acc[VEC + e] += g * (xv[e] - mu[e]) * is[e];
      }
    }
  }
  __shared__ float sm[2 * VEC * 256];
  col_block_reduce<2 * VEC>(acc, sw, rp, rr, sm);
  if (active && rr == 0) {
    float* out = partial + ((size_t)b * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      out[cv * VEC + e] = acc[e];
      out[C + cv * VEC + e] = acc[VEC + e];
    }
  }
}

static inline int se_bns_blocks_x(int B, long HW, int CV) {
  const ColLayout l = col_layout(CV);
  return col_blocks_x(HW, l, 4, 2048, B);
}
extern "C" int tfpp_se_bwd_apply_bns_rows(int B, int HW, int C, int dtype) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (B < 1 || HW < 1 || C % VEC) return TFPP_EINVAL;
  return B * se_bns_blocks_x(B, HW, C / VEC);
}
extern "C" int tfpp_se_bwd_apply_bns(const void* dy, const float* gate, const float* dpool, const void* y, const void* x, const float* save_mean,
                                     const float* save_invstd, void* dx, float* partial, int B, int HW, int C, int dtype, void* stream) {
  if (!dy || !gate || !dpool || !y || !x || !save_mean || !save_invstd || !dx || !partial) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC) return TFPP_EINVAL;
  const ColLayout l = col_layout(C / VEC);
  dim3 grid((unsigned)se_bns_blocks_x(B, HW, C / VEC), (unsigned)l.ny, (unsigned)B);
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(se_bwd_apply_bns_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, gate, dpool, (const float*)y, (const float*)x, save_mean,
                       save_invstd, (float*)dx, partial, (long)HW, C, l.sw, l.rp);
  else
    hipLaunchKernelGGL(se_bwd_apply_bns_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, gate, dpool, (const bf16_t*)y, (const bf16_t*)x,
                       save_mean, save_invstd, (bf16_t*)dx, partial, (long)HW, C, l.sw, l.rp);
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

// s = a + dropout(b) (written: LayerNorm's backward reads it) and y = LayerNorm(s) in one pass: the post-norm residual step of
// nn.TransformerDecoderLayer (x = norm(x + dropout(sublayer(x)))).  Dropout: hash, seed and element index (row * C + c) of tfpp_add_dropout.
template <typename T, int LN_MAXV>
__global__ void add_layernorm_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ sum, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                         long rows, int C, float eps, float p, float inv_keep, unsigned long long seed,
                                         const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;
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
      float bb[VEC];
      load_vec<T>(a + (size_t)row * C + cv * VEC, v[k]);
      load_vec<T>(b + (size_t)row * C + cv * VEC, bb);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if (p > 0.f) bb[e] *= dropout_scale(seed, (unsigned long long)row * C + cv * VEC + e, p, inv_keep);
        // round the sum to the storage type first: the statistics are those of the tensor the backward (and the unfused path) sees
        v[k][e] = ElemTraits<T>::to_f(ElemTraits<T>::from_f(bb[e] + v[k][e]));
        s += v[k][e];
      }
      store_vec<T>(sum + (size_t)row * C + cv * VEC, v[k]);
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

extern "C" int tfpp_add_layernorm_fwd(const void* a, const void* b, void* sum, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                      int64_t rows, int C, float eps, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype, void* stream) {
  if (!a || !b || !sum || !y || !gamma || !beta || p_drop < 0.f || p_drop >= 1.f) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC || C / VEC > 64 * 6) return TFPP_EINVAL;
  const int nv = (C / VEC + 63) / 64;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  dim3 grid((unsigned)((rows + 3) / 4));
#define ALN_FWD(TT, MV) hipLaunchKernelGGL((add_layernorm_fwd_kernel<TT, MV>), grid, dim3(256), 0, st, (const TT*)a, (const TT*)b, (TT*)sum, gamma, beta, (TT*)y, mean, rstd, (long)rows, C, eps, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset)
#define ALN_FWD_T(TT) do { if (nv <= 1) ALN_FWD(TT, 1); else if (nv <= 2) ALN_FWD(TT, 2); else if (nv <= 3) ALN_FWD(TT, 3); else if (nv <= 4) ALN_FWD(TT, 4); else ALN_FWD(TT, 6); } while (0)
  if (dtype == TFPP_F32) ALN_FWD_T(float); else ALN_FWD_T(bf16_t);
  TFPP_CHECK_LAUNCH();
  return 0;
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
// dx only: the parameter gradients are a separate column reduction (layernorm_param_grad_kernel), which keeps this kernel's
// live state to the row itself (g, xhat, gamma) instead of two more per-channel accumulators.
template <typename T, int LN_MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx, long rows, int C,
                                     int rows_per_wave, T* __restrict__ dx2, float p, float inv_keep,
                                     unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  // dx2 (nullable): dx times the dropout mask of (seed, row * C + c) -- the gradient of b in  LayerNorm(a + dropout(b))
  if (dx2 && seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long r0 = w * rows_per_wave;
  for (long row = r0; row < r0 + rows_per_wave && row < rows; ++row) {
    const float mu = mean[row], rs = rstd[row];
    float g[LN_MAXV][VEC], xh[LN_MAXV][VEC];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int cv = lane + k * 64;
      if (cv < CV) {
        float v[VEC], gm[VEC];
        load_vec<T>(dy + (size_t)row * C + cv * VEC, g[k]);
        load_vec<T>(x + (size_t)row * C + cv * VEC, v);
        load_vec<float>(gamma + cv * VEC, gm);
        if (VEC == 8) load_vec<float>(gamma + cv * VEC + 4, gm + 4);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          xh[k][e] = (v[e] - mu) * rs;
          g[k][e] *= gm[e];
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
        if (dx2) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)  // (the mask applies to the ROUNDED dx, as the separate add_dropout launch would see it)
            o[e] = ElemTraits<T>::to_f(ElemTraits<T>::from_f(o[e])) * (p > 0.f ? dropout_scale(seed, (unsigned long long)row * C + cv * VEC + e, p, inv_keep) : 1.f);
          store_vec<T>(dx2 + (size_t)row * C + cv * VEC, o);
        }
      }
    }
  }
}

// parameter gradients of LayerNorm as a column reduction: dgamma[c] += sum_r dy*xhat, dbeta[c] += sum_r dy.
// grid (ceil(C/64), S); block = 64 channels x 4 row slots.  The S row blocks of a channel block publish their partial sums and the one that
// draws the last ticket adds them in row-block order (fixed-order grid sum, common.h): bit-reproducible, one launch.
template <typename T>
__global__ void layernorm_param_grad_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                            const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                            long rows, int C, long rows_per_block, float* __restrict__ scratch) {
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
  // partials: [gridDim.y][C] pairs (dgamma, dbeta) behind the tickets -- one 8-byte agent-scope access per channel; ticket blockIdx.x counts the
  // row blocks of this channel block.  The last block adds them with all four row slots in parallel (slot s takes row blocks s, s + 4, ...:
  // agent-scope loads are issued one after the other, 16 instead of 64 round trips), then the four slot sums in slot order: a fixed order.
  unsigned long long* part = reinterpret_cast<unsigned long long*>(scratch + TFPP_GRIDSUM_TICKETS);
  if (slot == 0 && c < C) {
    const float tg = sm[0][0][cl] + sm[0][1][cl] + sm[0][2][cl] + sm[0][3][cl], tb = sm[1][0][cl] + sm[1][1][cl] + sm[1][2][cl] + sm[1][3][cl];
    const unsigned long long pk = (unsigned long long)__float_as_uint(tg) | ((unsigned long long)__float_as_uint(tb) << 32);
    __hip_atomic_store(part + (size_t)blockIdx.y * C + c, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!grid_last_ticket(reinterpret_cast<unsigned*>(scratch) + blockIdx.x, gridDim.y)) return;
  float tg = 0.f, tb = 0.f;
  if (c < C && slot < (int)gridDim.y)  // gridDim.y <= 64: at most 16 row blocks per slot, fetched in ONE round trip (common.h grid_fetch_pair_sum16)
    grid_fetch_pair_sum16(part + (size_t)slot * C + c, 4l * C, ((int)gridDim.y - slot + 3) / 4, tg, tb);
  sm[0][slot][cl] = tg;
  sm[1][slot][cl] = tb;
  __syncthreads();
  if (slot == 0 && c < C) {
    if (dgamma) dgamma[c] += sm[0][0][cl] + sm[0][1][cl] + sm[0][2][cl] + sm[0][3][cl];
    if (dbeta) dbeta[c] += sm[1][0][cl] + sm[1][1][cl] + sm[1][2][cl] + sm[1][3][cl];
  }
}

static int launch_layernorm_param_grad(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma, float* dbeta, int64_t rows,
                                       int C, int dtype, float* scratch, hipStream_t st) {
  if (!scratch || C > 3072) return TFPP_EINVAL;
  long S = rows / 16;
  if (S > 64) S = 64;
  if (S < 1) S = 1;
  const long rpb = (rows + S - 1) / S;
  dim3 g2((unsigned)((C + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
  if (dtype == TFPP_F32) hipLaunchKernelGGL(layernorm_param_grad_kernel<float>, g2, dim3(256), 0, st, (const float*)dy, (const float*)x, mean, rstd, dgamma, dbeta, (long)rows, C, rpb, scratch);
  else hipLaunchKernelGGL(layernorm_param_grad_kernel<bf16_t>, g2, dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, mean, rstd, dgamma, dbeta, (long)rows, C, rpb, scratch);
  TFPP_CHECK_LAUNCH();
  return 0;
}

static int launch_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, void* dx2, float p_drop,
                                uint64_t seed, const uint64_t* seed_offset, float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int dtype, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dx || p_drop < 0.f || p_drop >= 1.f || ((dgamma || dbeta) && !scratch)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (C % VEC || C / VEC > 64 * 6) return TFPP_EINVAL;
  const int nv = (C / VEC + 63) / 64;
  const int rpw = 1;  // dx: one wave per row, no atomics (parameter gradients come from the column-reduction kernel)
  const long waves = (rows + rpw - 1) / rpw;
  dim3 grid((unsigned)((waves + 3) / 4));
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
#define LN_BWD(TT, MV) hipLaunchKernelGGL((layernorm_bwd_kernel<TT, MV>), grid, dim3(256), 0, st, (const TT*)dy, (const TT*)x, gamma, mean, rstd, (TT*)dx, (long)rows, C, rpw, (TT*)dx2, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset)
#define LN_BWD_T(TT) do { if (nv <= 1) LN_BWD(TT, 1); else if (nv <= 2) LN_BWD(TT, 2); else if (nv <= 3) LN_BWD(TT, 3); else if (nv <= 4) LN_BWD(TT, 4); else LN_BWD(TT, 6); } while (0)
  if (dtype == TFPP_F32) LN_BWD_T(float); else LN_BWD_T(bf16_t);
  TFPP_CHECK_LAUNCH();
  if (dgamma || dbeta) return launch_layernorm_param_grad(dy, x, mean, rstd, dgamma, dbeta, rows, C, dtype, scratch, st);
  return 0;
}

extern "C" int tfpp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                                  float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int dtype, void* stream) {
  return launch_layernorm_bwd(dy, x, gamma, mean, rstd, dx, nullptr, 0.f, 0, nullptr, dgamma, dbeta, scratch, rows, C, dtype, stream);
}

// backward of tfpp_add_layernorm_fwd: d_sum (the gradient of a) and d_b = d_sum * dropout mask in one launch; parameter gradients as above
extern "C" int tfpp_add_layernorm_bwd(const void* dy, const void* sum, const float* gamma, const float* mean, const float* rstd, void* d_sum, void* d_b,
                                      float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, float p_drop, uint64_t seed,
                                      const uint64_t* seed_offset, int dtype, void* stream) {
  if (!d_b) return TFPP_EINVAL;
  return launch_layernorm_bwd(dy, sum, gamma, mean, rstd, d_sum, d_b, p_drop, seed, seed_offset, dgamma, dbeta, scratch, rows, C, dtype, stream);
}

// the parameter gradients alone (dgamma[c] += sum_r dy * xhat, dbeta[c] += sum_r dy): they only feed the optimizer, so the engine runs them
// on the weight-gradient lane and keeps the dx kernel alone on the dY chain
extern "C" int tfpp_layernorm_param_grad(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* scratch,
                                         int64_t rows, int C, int dtype, void* stream) {
  if (!dy || !x || !mean || !rstd || (!dgamma && !dbeta)) return TFPP_EINVAL;
  return launch_layernorm_param_grad(dy, x, mean, rstd, dgamma, dbeta, rows, C, dtype, scratch, (hipStream_t)stream);
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
