// Train-mode BatchNorm passes with the statistics step folded into the pass that needs it (round 6).
//
// Rounds 1-5 ran conv (statistics in the epilogue) -> bn_finalize_partials -> affine_act forward and bn_reduce -> bn_bwd_coef -> bn_bwd_apply
// backward: the two middle kernels are a few KB of work behind a dependent launch each (136 + 136 launches, ~1 ms of kernel time and, on the
// chain of the captured step, 14-20 us per launch including the gap in front of it).  Here the kernel that CONSUMES the statistics adds the
// per-M-tile rows itself, in its prologue, beside its own first loads:
//   * thread layout "channel block": a workgroup owns CB 16-byte chunk columns (~64 channels) x RS row slots, so it needs the rows of ~64
//     channels only (nrows x 0.5 KB; with the column-fixed layout of the older kernels -- all C channels per workgroup -- every workgroup would
//     read the whole nrows x 2C table);
//   * the sums run in double in a fixed order (row group by row group, then the groups in index order): bit-reproducible;
//   * workgroup (row block 0) of every channel block writes what later kernels read: scale / shift / saved mean / invstd, running statistics
//     (forward), dgamma / dbeta (backward).
// The same layout carries the squeeze-excite passes around a conv2 output that is never normalised in memory (mask and activation are
// recomputed from the raw tensor: y = relu(raw * scale + shift)).
#include "bn_rows.h"

namespace {

// ---- channel-block layout ------------------------------------------------------------------------------------------------------------------
struct CbLayout { int cb, rs, ncb; };  // chunk columns per block, row slots (rs * cb <= 256), channel blocks
inline CbLayout cb_layout(int CV) {
  CbLayout l;
  if (CV <= 12) l.cb = CV;
  else if (CV % 8 == 0) l.cb = 8;
  else if (CV % 9 == 0) l.cb = 9;
  else if (CV % 7 == 0) l.cb = 7;
  else if (CV % 10 == 0) l.cb = 10;
  else if (CV % 6 == 0) l.cb = 6;
  else l.cb = 8;  // last block partly idle
  l.rs = 256 / l.cb;
  l.ncb = (CV + l.cb - 1) / l.cb;
  return l;
}
// rows per thread so that a tensor of `rows` rows gives at most max_blocks row blocks
inline int cb_rows_per_thread(long rows, int rs, int min_rpt, long max_blocks) {
  long rpt = min_rpt;
  while ((rows + (long)rs * rpt - 1) / ((long)rs * rpt) > max_blocks) ++rpt;
  return (int)rpt;
}

__device__ __forceinline__ bool cb_thread(int cb, int rs, int CV, int& rr, int& cv) {
  rr = (int)threadIdx.x / cb;
  cv = (int)blockIdx.y * cb + ((int)threadIdx.x - rr * cb);
  return rr < rs && cv < CV;
}

// ---- forward: y = f(BN(x)) -----------------------------------------------------------------------------------------------------------------
template <typename T, bool RES, bool GATE>
__global__ __launch_bounds__(256, 4) void bn_apply_rows_kernel(const T* __restrict__ x, tfpp_bn_rows bn, const T* __restrict__ res,
                                                            const float* __restrict__ gate, T* __restrict__ y, long rows, long rows_per_batch,
                                                            int relu_pre, int relu_post, int cb, int rs, int rpt) {
  constexpr int VEC = ElemTraits<T>::VEC, U = 4;
  const int C = bn.C, CV = C / VEC;
  __shared__ double sm[512];
  __shared__ float sc_s[96], sh_s[96];
  int rr, cv;
  const bool active = cb_thread(cb, rs, CV, rr, cv);
  const int c0 = cv * VEC;
  const long r0 = (long)blockIdx.x * rs * rpt + rr;
  // first rows of this thread: in flight while the prologue runs
  uint4 xv[U], rv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long r = r0 + (long)u * rs;
    xv[u] = make_uint4(0, 0, 0, 0);
    rv[u] = make_uint4(0, 0, 0, 0);
    if (active && u < rpt && r < rows) {
      xv[u] = *reinterpret_cast<const uint4*>(x + (size_t)r * C + c0);
      if (RES) rv[u] = *reinterpret_cast<const uint4*>(res + (size_t)r * C + c0);
    }
  }
  bn_block_scale_shift(bn, (int)blockIdx.y * cb * VEC, cb * VEC, blockIdx.x == 0, blockIdx.x == 0 && blockIdx.y == 0, sm, sc_s, sh_s);
  if (!active) return;
  float sc[VEC], sh[VEC];
  const int cl = ((int)threadIdx.x - rr * cb) * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = sc_s[cl + e]; sh[e] = sh_s[cl + e]; }
  auto body = [&](long r, const uint4& xq, const uint4& rq) {
    float v[VEC], q[VEC];
    unpack16<T>(xq, v);
    if (RES) unpack16<T>(rq, q);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = v[e] * sc[e] + sh[e];
      if (relu_pre) t = t > 0.f ? t : 0.f;
      v[e] = t;
    }
    if (GATE) {
      const float4* gp = reinterpret_cast<const float4*>(gate + (size_t)((unsigned long)r / (unsigned long)rows_per_batch) * C + c0);
#pragma unroll
      for (int h = 0; h < VEC / 4; ++h) {
        const float4 g4 = gp[h];
        v[4 * h + 0] *= g4.x; v[4 * h + 1] *= g4.y; v[4 * h + 2] *= g4.z; v[4 * h + 3] *= g4.w;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = v[e];
      if (RES) t += q[e];
      if (relu_post) t = t > 0.f ? t : 0.f;
      v[e] = t;
    }
    store_vec<T>(y + (size_t)r * C + c0, v);
  };
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long r = r0 + (long)u * rs;
    if (u < rpt && r < rows) body(r, xv[u], rv[u]);
  }
  for (int i = U; i < rpt; i += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + (long)(i + u) * rs;
      if (i + u < rpt && r < rows) {
        xv[u] = *reinterpret_cast<const uint4*>(x + (size_t)r * C + c0);
        if (RES) rv[u] = *reinterpret_cast<const uint4*>(res + (size_t)r * C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + (long)(i + u) * rs;
      if (i + u < rpt && r < rows) body(r, xv[u], rv[u]);
    }
  }
}

template <typename T>
int launch_bn_apply_rows(const void* x, const tfpp_bn_rows& bn, const void* res, const float* gate, void* y, long rows, long rows_per_batch,
                         int relu_pre, int relu_post, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (bn.C % VEC) return TFPP_EINVAL;
  const CbLayout l = cb_layout(bn.C / VEC);
  const int rpt = cb_rows_per_thread(rows, l.rs, 4, 1 << 20);
  dim3 grid((unsigned)((rows + (long)l.rs * rpt - 1) / ((long)l.rs * rpt)), (unsigned)l.ncb);
#define BA(R_, G_) hipLaunchKernelGGL((bn_apply_rows_kernel<T, R_, G_>), grid, dim3(256), 0, st, (const T*)x, bn, (const T*)res, gate, (T*)y, rows, rows_per_batch, relu_pre, relu_post, l.cb, l.rs, rpt)
  if (res) { if (gate) BA(true, true); else BA(true, false); }
  else { if (gate) BA(false, true); else BA(false, false); }
#undef BA
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---- backward ------------------------------------------------------------------------------------------------------------------------------
// mask: 0 none, 1 y > 0 (forward output), 2 x * scale + shift > 0 (recomputed from the raw tensor)
template <typename T, int MASK>
__device__ __forceinline__ void masked_grad(const uint4& gq, const uint4& yq, const float* xv, const float* sc, const float* sh, float* g) {
  constexpr int VEC = ElemTraits<T>::VEC;
  unpack16<T>(gq, g);
  if (MASK == 1) {
    float o[VEC];
    unpack16<T>(yq, o);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
  } else if (MASK == 2) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] = (xv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
  }
}

// Sum NV per-thread accumulators (element e of chunk column cc at acc[e], tid = rr * cb + cc) over the rs row slots of a channel-block workgroup
// in ONE step: every thread parks its values in LDS, then thread t < NV * cb adds the rs addends of one (value, column) pair in row-slot order
// (fixed order: deterministic).  Returns true in those threads; `tot` is the sum of value index e = t / cb ... see the callers' mapping:
// t -> (h = t / nch, ch = t % nch) with nch = cb * VEC, value index e = h * VEC + ch % VEC, column cc = ch / VEC.
// (The first version was a 5-level tree with a barrier per level: ~1.5 us per workgroup for rs = 32.)  sm: NV * SM_PITCH floats.
#define CB_SM_PITCH 257
template <int NV, int VEC> __device__ __forceinline__ bool cb_block_sum(const float (&acc)[NV], int cb, int rs, float* sm, int& h, int& ch, float& tot) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < NV; ++e) sm[e * CB_SM_PITCH + tid] = acc[e];
  __syncthreads();
  const int nch = cb * VEC;
  h = tid / nch;
  ch = tid - h * nch;
  if (h >= NV / VEC) return false;
  const int e = h * VEC + ch % VEC, cc = ch / VEC;
  const float* p = sm + e * CB_SM_PITCH + cc;
  float t0 = 0.f;
  for (int rr = 0; rr < rs; ++rr) t0 += p[rr * cb];
  tot = t0;
  return true;
}

template <typename T, int MASK>
__global__ __launch_bounds__(256, 4) void bn_bwd_reduce_rows_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 float* __restrict__ partial, long rows, int C, int cb, int rs, int rpt) {
  constexpr int VEC = ElemTraits<T>::VEC, U = 4;
  const int CV = C / VEC;
  int rr, cv;
  const bool active = cb_thread(cb, rs, CV, rr, cv);
  const int c0 = cv * VEC;
  float acc[2 * VEC];
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) acc[e] = 0.f;
  if (active) {
    float mu[VEC], is[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
      sc[e] = MASK == 2 ? scale[c0 + e] : 1.f; sh[e] = MASK == 2 ? shift[c0 + e] : 0.f;
    }
    const long r0 = (long)blockIdx.x * rs * rpt + rr;
    for (int i = 0; i < rpt; i += U) {
      uint4 gq[U], xq[U], yq[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = r0 + (long)(i + u) * rs;
        ok[u] = i + u < rpt && r < rows;
        gq[u] = xq[u] = yq[u] = make_uint4(0, 0, 0, 0);
        if (ok[u]) {
          const size_t off = (size_t)r * C + c0;
          gq[u] = *reinterpret_cast<const uint4*>(dy + off);
          xq[u] = *reinterpret_cast<const uint4*>(x + off);
          if (MASK == 1) yq[u] = *reinterpret_cast<const uint4*>(y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float g[VEC], xv[VEC];
        unpack16<T>(xq[u], xv);
        masked_grad<T, MASK>(gq[u], yq[u], xv, sc, sh, g);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc[e] += g[e]; acc[VEC + e] += g[e] * (xv[e] - mu[e]) * is[e]; }
      }
    }
  }
  __shared__ float sm[2 * VEC * CB_SM_PITCH];
  int h, ch;
  float tot;
  if (cb_block_sum<2 * VEC, VEC>(acc, cb, rs, sm, h, ch, tot)) {
    const int c = (int)blockIdx.y * cb * VEC + ch;
    if (c < C) partial[(size_t)blockIdx.x * 2 * C + (size_t)h * C + c] = tot;
  }
}

template <typename T, int MASK, bool DRES>
__global__ __launch_bounds__(256, 4) void bn_bwd_apply_rows_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ x,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ partial, int nrows,
                                                                T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, long rows, int C, int cb, int rs, int rpt) {
  constexpr int VEC = ElemTraits<T>::VEC, U = (MASK == 1 || DRES) ? 2 : 4;  // (three operand streams / two result streams: 4 rows in flight spill)
  const int CV = C / VEC;
  __shared__ double sm[512];
  __shared__ float ka_s[96], kb_s[96], kd_s[96], sc_s[96], sh_s[96];
  int rr, cv;
  const bool active = cb_thread(cb, rs, CV, rr, cv);
  const int c0 = cv * VEC;
  const long r0 = (long)blockIdx.x * rs * rpt + rr;
  uint4 gq[U], xq[U], yq[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long r = r0 + (long)u * rs;
    gq[u] = xq[u] = yq[u] = make_uint4(0, 0, 0, 0);
    if (active && u < rpt && r < rows) {
      const size_t off = (size_t)r * C + c0;
      gq[u] = *reinterpret_cast<const uint4*>(dy + off);
      xq[u] = *reinterpret_cast<const uint4*>(x + off);
      if (MASK == 1) yq[u] = *reinterpret_cast<const uint4*>(y + off);
    }
  }
  {  // coefficients of this channel block: dx = A g + Bc x + D  (bn_bwd_coef_kernel's formulas)
    const int c_base = (int)blockIdx.y * cb * VEC, nch = cb * VEC, t = threadIdx.x;
    int c;
    double s0, s1;
    if (rows_block_sum<16>(partial, nrows, C, c_base, nch, sm, c, s0, s1)) {  // (16: with the three operand streams of the main loop 24 spills)
      const double n = (double)rows;
      const double gm = gamma ? (double)gamma[c] : 1.0, is = (double)invstd[c], mu = (double)mean[c];
      const double A = gm * is, Bc = -gm * is * is * s1 / n, D = -gm * is * s0 / n - Bc * mu;
      ka_s[t] = (float)A;
      kb_s[t] = (float)Bc;
      kd_s[t] = (float)D;
      if (MASK == 2) { sc_s[t] = scale[c]; sh_s[t] = shift[c]; }
      if (blockIdx.x == 0) {
        if (dgamma) dgamma[c] += (float)s1;
        if (dbeta) dbeta[c] += (float)s0;
      }
    }
    __syncthreads();
  }
  if (!active) return;
  float ka[VEC], kb[VEC], kd[VEC], sc[VEC], sh[VEC];
  const int cl = ((int)threadIdx.x - rr * cb) * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    ka[e] = ka_s[cl + e]; kb[e] = kb_s[cl + e]; kd[e] = kd_s[cl + e];
    sc[e] = MASK == 2 ? sc_s[cl + e] : 1.f; sh[e] = MASK == 2 ? sh_s[cl + e] : 0.f;
  }
  auto body = [&](long r, const uint4& g4, const uint4& x4, const uint4& y4) {
    float g[VEC], v[VEC];
    unpack16<T>(x4, v);
    masked_grad<T, MASK>(g4, y4, v, sc, sh, g);
    const size_t off = (size_t)r * C + c0;
    if (DRES) store_vec<T>(dres + off, g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = ka[e] * g[e] + kb[e] * v[e] + kd[e];
    store_vec<T>(dx + off, v);
  };
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long r = r0 + (long)u * rs;
    if (u < rpt && r < rows) body(r, gq[u], xq[u], yq[u]);
  }
  for (int i = U; i < rpt; i += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + (long)(i + u) * rs;
      if (i + u < rpt && r < rows) {
        const size_t off = (size_t)r * C + c0;
        gq[u] = *reinterpret_cast<const uint4*>(dy + off);
        xq[u] = *reinterpret_cast<const uint4*>(x + off);
        if (MASK == 1) yq[u] = *reinterpret_cast<const uint4*>(y + off);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + (long)(i + u) * rs;
      if (i + u < rpt && r < rows) body(r, gq[u], xq[u], yq[u]);
    }
  }
}

// row blocks (= partial rows) and rows per thread of the backward reduce for a [rows, C] tensor: at most TFPP_BN_ROWS_MAX / 2 row blocks
inline void bwd_geometry(long rows, int CV, CbLayout& l, int& rpt, int& nblk) {
  l = cb_layout(CV);
  rpt = cb_rows_per_thread(rows, l.rs, 4, TFPP_BN_ROWS_MAX / 2);
  nblk = (int)((rows + (long)l.rs * rpt - 1) / ((long)l.rs * rpt));
}

template <typename T>
int launch_bn_bwd_reduce_rows(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* mean,
                              const float* invstd, float* partial, long rows, int C, int mask, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  CbLayout l;
  int rpt, nblk;
  bwd_geometry(rows, C / VEC, l, rpt, nblk);
  dim3 grid((unsigned)nblk, (unsigned)l.ncb);
#define BR(M_) hipLaunchKernelGGL((bn_bwd_reduce_rows_kernel<T, M_>), grid, dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, scale, shift, mean, invstd, partial, rows, C, l.cb, l.rs, rpt)
  if (mask == 0) BR(0); else if (mask == 1) BR(1); else BR(2);
#undef BR
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <typename T>
int launch_bn_bwd_apply_rows(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* gamma,
                             const float* mean, const float* invstd, const float* partial, int nrows, void* dx, void* dres, float* dgamma,
                             float* dbeta, long rows, int C, int mask, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  const CbLayout l = cb_layout(C / VEC);
  const int rpt = cb_rows_per_thread(rows, l.rs, 4, 1 << 20);
  dim3 grid((unsigned)((rows + (long)l.rs * rpt - 1) / ((long)l.rs * rpt)), (unsigned)l.ncb);
#define BA(M_, D_) hipLaunchKernelGGL((bn_bwd_apply_rows_kernel<T, M_, D_>), grid, dim3(256), 0, st, (const T*)dy, (const T*)y, (const T*)x, scale, shift, gamma, mean, invstd, partial, nrows, (T*)dx, (T*)dres, dgamma, dbeta, rows, C, l.cb, l.rs, rpt)
#define BA_M(M_) do { if (dres) BA(M_, true); else BA(M_, false); } while (0)
  if (mask == 0) BA_M(0); else if (mask == 1) BA_M(1); else BA_M(2);
#undef BA_M
#undef BA
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---- squeeze-excite passes around a tensor that exists only as (raw, statistics) -----------------------------------------------------------
// pool: out[b][c] = mul * sum_hw relu(x * scale + shift)  (the squeeze of a conv2 output that is never normalised in memory).
// One launch: the (<= 16) row-block workgroups of a sample publish their partial sums and draw a ticket per sample, the last one adds them in
// row-block order (common.h).  grid = (row blocks per sample, channel blocks, B).
template <typename T>
__global__ __launch_bounds__(256, 4) void hw_reduce_bn_kernel(const T* __restrict__ x, tfpp_bn_rows bn,
                                                           float* __restrict__ partial, float* __restrict__ out, unsigned* __restrict__ tickets,
                                                           int HW, int cb, int rs, int rpt, float mul) {
  constexpr int VEC = ElemTraits<T>::VEC, U = 4;
  const int C = bn.C, CV = C / VEC;
  __shared__ double smd[512];
  __shared__ float sc_s[96], sh_s[96];
  __shared__ float sm[VEC * CB_SM_PITCH];
  int rr, cv;
  const bool active = cb_thread(cb, rs, CV, rr, cv);
  const int c0 = cv * VEC, b = blockIdx.z;
  const long r0 = (long)blockIdx.x * rs * rpt + rr;
  const T* xb = x + (size_t)b * HW * C;
  uint4 xq[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long r = r0 + (long)u * rs;
    xq[u] = make_uint4(0, 0, 0, 0);
    if (active && u < rpt && r < HW) xq[u] = *reinterpret_cast<const uint4*>(xb + (size_t)r * C + c0);
  }
  bn_block_scale_shift(bn, (int)blockIdx.y * cb * VEC, cb * VEC, blockIdx.x == 0 && b == 0, blockIdx.x == 0 && blockIdx.y == 0 && b == 0, smd, sc_s,
                       sh_s);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (active) {
    float sc[VEC], sh[VEC];
    const int cl = ((int)threadIdx.x - rr * cb) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { sc[e] = sc_s[cl + e]; sh[e] = sh_s[cl + e]; }
    auto body = [&](const uint4& x4) {
      float v[VEC];
      unpack16<T>(x4, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float t = v[e] * sc[e] + sh[e];
        acc[e] += t > 0.f ? t : 0.f;
      }
    };
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long r = r0 + (long)u * rs;
      if (u < rpt && r < HW) body(xq[u]);
    }
    for (int i = U; i < rpt; i += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = r0 + (long)(i + u) * rs;
        if (i + u < rpt && r < HW) xq[u] = *reinterpret_cast<const uint4*>(xb + (size_t)r * C + c0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = r0 + (long)(i + u) * rs;
        if (i + u < rpt && r < HW) body(xq[u]);
      }
    }
  }
  {
    int h, ch;
    float tot;
    if (cb_block_sum<VEC, VEC>(acc, cb, rs, sm, h, ch, tot)) {
      const int c = (int)blockIdx.y * cb * VEC + ch;
      if (c < C) grid_publish(partial + ((size_t)b * gridDim.x + blockIdx.x) * C + c, tot);
    }
  }
  if (!grid_last_ticket(tickets + b, gridDim.x * gridDim.y)) return;
  for (int c = threadIdx.x; c < C; c += 256)
    out[(size_t)b * C + c] = grid_fetch_sum16(partial + (size_t)b * gridDim.x * C + c, C, (int)gridDim.x) * mul;
}

template <typename T>
int launch_hw_reduce_bn(const void* x, const tfpp_bn_rows& bn, float* out, float* scratch, float* tickets, int B, int HW, float mulv,
                        hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (bn.C % VEC || B < 1 || B > TFPP_GRIDSUM_TICKETS) return TFPP_EINVAL;
  const CbLayout l = cb_layout(bn.C / VEC);
  const int rpt = cb_rows_per_thread(HW, l.rs, 4, 16);  // <= 16 row blocks per sample: one batch of agent-scope loads per channel in the tail
  dim3 grid((unsigned)((HW + l.rs * rpt - 1) / (l.rs * rpt)), (unsigned)l.ncb, (unsigned)B);
  hipLaunchKernelGGL((hw_reduce_bn_kernel<T>), grid, dim3(256), 0, st, (const T*)x, bn, scratch, out, reinterpret_cast<unsigned*>(tickets), HW, l.cb, l.rs, rpt, mulv);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dx = dy * gate[b,c] + dpool[b,c] / HW  (complete gradient of a2 = relu(BN2(raw2))), and one row of (sum g, sum g*xhat) per workgroup with
// g = dx (rounded) * (raw * scale + shift > 0).  grid = (row blocks per sample, channel blocks, B); partial row = b * gridDim.x + blockIdx.x.
template <typename T>
__global__ __launch_bounds__(256, 4) void se_bwd_apply_bn_kernel(const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool,
                                                              const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, T* __restrict__ dx,
                                                              float* __restrict__ partial, int HW, int C, int cb, int rs, int rpt) {
  constexpr int VEC = ElemTraits<T>::VEC, U = 2;  // (six per-channel parameter vectors live in registers: 4 rows in flight spill)
  const int CV = C / VEC;
  int rr, cv;
  const bool active = cb_thread(cb, rs, CV, rr, cv);
  const int b = blockIdx.z, c0 = cv * VEC;
  float acc[2 * VEC];
#pragma unroll
  for (int e = 0; e < 2 * VEC; ++e) acc[e] = 0.f;
  if (active) {
    float gt[VEC], dp[VEC], mu[VEC], is[VEC], sc[VEC], sh[VEC];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      gt[e] = gate[(size_t)b * C + c0 + e];
      dp[e] = dpool[(size_t)b * C + c0 + e] * inv;
      mu[e] = mean[c0 + e]; is[e] = invstd[c0 + e];
      sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e];
    }
    const long r0 = (long)blockIdx.x * rs * rpt + rr;
    for (int i = 0; i < rpt; i += U) {
      uint4 gq[U], xq[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long r = r0 + (long)(i + u) * rs;
        ok[u] = i + u < rpt && r < HW;
        gq[u] = xq[u] = make_uint4(0, 0, 0, 0);
        if (ok[u]) {
          const size_t off = ((size_t)b * HW + r) * C + c0;
          gq[u] = *reinterpret_cast<const uint4*>(dy + off);
          xq[u] = *reinterpret_cast<const uint4*>(x + off);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        const size_t off = ((size_t)b * HW + r0 + (long)(i + u) * rs) * C + c0;
        float v[VEC], xv[VEC];
        unpack16<T>(gq[u], v);
        unpack16<T>(xq[u], xv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = v[e] * gt[e] + dp[e];
        const uint4 packed = pack16<T>(v);
        *reinterpret_cast<uint4*>(dx + off) = packed;
        unpack16<T>(packed, v);  // the rounded values the BatchNorm backward reads back
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float g = (xv[e] * sc[e] + sh[e]) > 0.f ? v[e] : 0.f;
          acc[e] += g;
          acc[VEC + e] += g * (xv[e] - mu[e]) * is[e];
        }
      }
    }
  }
  __shared__ float sm[2 * VEC * CB_SM_PITCH];
  int h, ch;
  float tot;
  if (cb_block_sum<2 * VEC, VEC>(acc, cb, rs, sm, h, ch, tot)) {
    const int c = (int)blockIdx.y * cb * VEC + ch;
    if (c < C) partial[((size_t)b * gridDim.x + blockIdx.x) * 2 * C + (size_t)h * C + c] = tot;
  }
}

inline void se_bn_geometry(int B, long HW, int CV, CbLayout& l, int& rpt, int& nb) {
  l = cb_layout(CV);
  long cap = (TFPP_BN_ROWS_MAX / 2) / B;
  if (cap < 1) cap = 1;
  rpt = cb_rows_per_thread(HW, l.rs, 4, cap);
  nb = (int)((HW + (long)l.rs * rpt - 1) / ((long)l.rs * rpt));
}
}  // namespace

extern "C" int tfpp_bn_apply_rows(const void* x, const tfpp_bn_rows* bn, const void* res, const float* gate, void* y, int64_t rows,
                                  int64_t rows_per_batch, int relu_pre, int relu_post, int dtype, void* stream) {
  if (!x || !bn || !y || !bn->scale || !bn->shift || rows < 1 || (gate && rows_per_batch < 1)) return TFPP_EINVAL;
  if (bn->partial && (bn->nrows < 1 || bn->count < 1)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_apply_rows<float>(x, *bn, res, gate, y, (long)rows, (long)rows_per_batch, relu_pre, relu_post, st)
                           : launch_bn_apply_rows<bf16_t>(x, *bn, res, gate, y, (long)rows, (long)rows_per_batch, relu_pre, relu_post, st);
}

extern "C" int tfpp_bn_bwd_rows_count(int64_t rows, int C, int dtype) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (rows < 1 || C < VEC || C % VEC) return TFPP_EINVAL;
  CbLayout l;
  int rpt, nblk;
  bwd_geometry((long)rows, C / VEC, l, rpt, nblk);
  return nblk;
}

extern "C" int tfpp_bn_bwd_reduce_rows(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* save_mean,
                                       const float* save_invstd, float* partial, int64_t rows, int C, int mask, int dtype, void* stream) {
  if (!dy || !x || !partial || !save_mean || !save_invstd || mask < 0 || mask > 2 || (mask == 1 && !y) || (mask == 2 && (!scale || !shift)))
    return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_bn_bwd_reduce_rows<float>(dy, y, x, scale, shift, save_mean, save_invstd, partial, (long)rows, C, mask, st)
                           : launch_bn_bwd_reduce_rows<bf16_t>(dy, y, x, scale, shift, save_mean, save_invstd, partial, (long)rows, C, mask, st);
}

extern "C" int tfpp_bn_bwd_apply_rows2(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* gamma,
                                       const float* save_mean, const float* save_invstd, const float* partial, int nrows, void* dx, void* dres,
                                       float* dgamma, float* dbeta, int64_t rows, int C, int mask, int dtype, void* stream) {
  if (!dy || !x || !dx || !partial || nrows < 1 || !save_mean || !save_invstd || mask < 0 || mask > 2 || (mask == 1 && !y) ||
      (mask == 2 && (!scale || !shift)))
    return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32
             ? launch_bn_bwd_apply_rows<float>(dy, y, x, scale, shift, gamma, save_mean, save_invstd, partial, nrows, dx, dres, dgamma, dbeta, (long)rows, C, mask, st)
             : launch_bn_bwd_apply_rows<bf16_t>(dy, y, x, scale, shift, gamma, save_mean, save_invstd, partial, nrows, dx, dres, dgamma, dbeta, (long)rows, C, mask, st);
}

extern "C" int tfpp_mean_hw_bn(const void* x, const tfpp_bn_rows* bn, float* out, float* scratch, float* ticket_scratch, int B, int HW, int dtype,
                               void* stream) {
  if (!x || !bn || !out || !scratch || !ticket_scratch || !bn->scale || !bn->shift || HW < 1) return TFPP_EINVAL;
  if (bn->partial && (bn->nrows < 1 || bn->count < 1)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_hw_reduce_bn<float>(x, *bn, out, scratch, ticket_scratch, B, HW, 1.f / (float)HW, st)
                           : launch_hw_reduce_bn<bf16_t>(x, *bn, out, scratch, ticket_scratch, B, HW, 1.f / (float)HW, st);
}

extern "C" int tfpp_se_bwd_apply_bn_rows(int B, int HW, int C, int dtype) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (B < 1 || HW < 1 || C < VEC || C % VEC) return TFPP_EINVAL;
  CbLayout l;
  int rpt, nb;
  se_bn_geometry(B, HW, C / VEC, l, rpt, nb);
  return B * nb;
}

extern "C" int tfpp_se_bwd_apply_bn(const void* dy, const float* gate, const float* dpool, const void* x, const float* scale, const float* shift,
                                    const float* save_mean, const float* save_invstd, void* dx, float* partial, int B, int HW, int C, int dtype,
                                    void* stream) {
  if (!dy || !gate || !dpool || !x || !scale || !shift || !save_mean || !save_invstd || !dx || !partial) return TFPP_EINVAL;
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (B < 1 || HW < 1 || C < VEC || C % VEC) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  CbLayout l;
  int rpt, nb;
  se_bn_geometry(B, HW, C / VEC, l, rpt, nb);
  dim3 grid((unsigned)nb, (unsigned)l.ncb, (unsigned)B);
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(se_bwd_apply_bn_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, gate, dpool, (const float*)x, scale, shift, save_mean,
                       save_invstd, (float*)dx, partial, HW, C, l.cb, l.rs, rpt);
  else
    hipLaunchKernelGGL(se_bwd_apply_bn_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, gate, dpool, (const bf16_t*)x, scale, shift,
                       save_mean, save_invstd, (bf16_t*)dx, partial, HW, C, l.cb, l.rs, rpt);
  TFPP_CHECK_LAUNCH();
  return 0;
}
