// Implicit-GEMM convolution (forward / data-gradient), weight-gradient and strided batched GEMM on MFMA (gfx950).
// See gemm_core.h for the fragment / LDS layouts and include/tfpp.h for the semantics of each entry point.
#include "gemm_core.h"
#include "gemm_internal.h"
#include <algorithm>
#include <vector>
#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------------------------------------------
// conv / linear forward and data gradient
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int BKT, bool BNS = false>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * TFPP_WAVE) void conv_gemm_kernel(tfpp_conv_params p) {
  using C = TileCfg<T, BM, BN, WM, WN, BKT>;
  constexpr int VEC = C::VEC, KV = C::KV, NT = C::NT, BK = C::BK;
  // one raw buffer: the A / B stages during the K loop, the per-wave epilogue strips afterwards
  constexpr size_t STAGE_BYTES = (size_t)(C::A_ELEMS_RM + C::B_ELEMS_RM) * sizeof(T);
  constexpr size_t STRIP_BYTES = (size_t)(NT / 64) * EpiStrip<C::FN>::FLOATS * sizeof(float);
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[STAGE_BYTES > STRIP_BYTES ? STAGE_BYTES : STRIP_BYTES];
  T* As = reinterpret_cast<T*>(smem_raw);
  T* Bs = As + C::A_ELEMS_RM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  // grid: x = N-tile (fastest), y = M-tile, z = group.  Consecutive workgroups share the same activation rows (A) and
  // workgroup b lands on XCD b % 8, so each XCD's L2 keeps a fixed subset of weight panels (B) resident.
  int g = blockIdx.z, split = 0;
  if (p.splitk > 1) { g = blockIdx.z / p.splitk; split = blockIdx.z - g * p.splitk; }  // z = (group, K slice)
  const int mtile = blockIdx.y;
  const int bm0 = mtile * BM, bn0 = blockIdx.x * BN;
  const int M = p.B * p.Hd * p.Wd, K = p.R * p.S * p.ks_g;
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src);
  const T* __restrict__ wk = reinterpret_cast<const T*>(p.w) + (size_t)g * p.n_g * K;

  constexpr int A_IT = (BM * KV) / NT;
  static_assert((BM * KV) % NT == 0, "A tile must divide evenly");
  constexpr int B_IT = (BN * KV + NT - 1) / NT;
  int a_b[A_IT], a_h0[A_IT], a_w0[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int v = tid + i * NT, row = v / KV, m = bm0 + row;
    if (m < M) {
      const int hw = p.Hd * p.Wd, b = m / hw, pix = m - b * hw, hd = pix / p.Wd, wd = pix - hd * p.Wd;
      a_b[i] = b;
      if (p.mode == 0) { a_h0[i] = hd * p.stride - p.pad; a_w0[i] = wd * p.stride - p.pad; }
      else { a_h0[i] = hd + p.pad; a_w0[i] = wd + p.pad; }
    } else {
      a_b[i] = -1; a_h0[i] = 0; a_w0[i] = 0;
    }
  }
  f32x4_t acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nkt = (K + BK - 1) / BK;
  int kt_beg = 0, kt_end = nkt;
  if (p.splitk > 1) {
    const int per = (nkt + p.splitk - 1) / p.splitk;
    kt_beg = split * per;
    kt_end = kt_beg + per < nkt ? kt_beg + per : nkt;
  }
  // Register-prefetched K loop (round 6): the global loads of tile kt + 1 are issued before the MFMAs of tile kt and land while they run; the
  // first version (load -> LDS store -> barrier -> MFMA -> barrier per tile) exposed one memory latency per 32 / 64 channels of K, which is most
  // of the time of the short-K launches this kernel serves (stage-1 convolutions, the fp32 planning head, every fp32 GEMM of the parity path).
  uint4 a_val[A_IT], b_val[B_IT];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * NT, row = v / KV, kc = v - row * KV, k0 = kt * BK + kc * VEC;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (a_b[i] >= 0 && k0 < K) {
        const int rs = k0 / p.ks_g, c = k0 - rs * p.ks_g, r = rs / p.S, s = rs - r * p.S;
        int hs, ws;
        bool ok;
        if (p.mode == 0) {
          hs = a_h0[i] + r; ws = a_w0[i] + s;
          ok = (hs >= 0) & (hs < p.Hs) & (ws >= 0) & (ws < p.Ws);
        } else {
          const int th = a_h0[i] - r, tw = a_w0[i] - s;
          hs = th / p.stride; ws = tw / p.stride;
          ok = (th >= 0) & (tw >= 0) & (hs * p.stride == th) & (ws * p.stride == tw) & (hs < p.Hs) & (ws < p.Ws);
        }
        if (ok) val = *reinterpret_cast<const uint4*>(src + ((size_t)(a_b[i] * p.Hs + hs) * p.Ws + ws) * p.src_ld + g * p.ks_g + c);
      }
      a_val[i] = val;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (v < BN * KV) {
        const int row = v / KV, kc = v - row * KV, k0 = kt * BK + kc * VEC, n = bn0 + row;
        if (n < p.n_g && k0 < K) val = *reinterpret_cast<const uint4*>(wk + (size_t)n * K + k0);
      }
      b_val[i] = val;
    }
  };
  if (kt_beg < kt_end) load_tile(kt_beg);
  for (int kt = kt_beg; kt < kt_end; ++kt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * NT, row = v / KV, kc = v - row * KV;
      *reinterpret_cast<uint4*>(&As[row * C::LDK + kc * VEC]) = a_val[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * NT;
      if (v < BN * KV) {
        const int row = v / KV, kc = v - row * KV;
        *reinterpret_cast<uint4*>(&Bs[row * C::LDK + kc * VEC]) = b_val[i];
      }
    }
    __syncthreads();
    if (kt + 1 < kt_end) load_tile(kt + 1);  // in flight during the MFMAs below
    tile_mma_step<C, T, false, false>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }

  if (p.splitk > 1) {  // raw fp32 slice -> workspace [split][M][G*n_g]; splitk_epilogue_kernel finishes the job
    const int ntot = p.G * p.n_g;
    float* __restrict__ wsp = p.splitk_ws + (size_t)split * M * ntot;
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < C::FN; ++j) {
          const int n = bn0 + wn * WN + j * 16 + (lane & 15);
          if (n < p.n_g) wsp[(size_t)m * ntot + g * p.n_g + n] = acc[i][j][r];
        }
      }
    return;
  }

  // optional fused BatchNorm statistics: per-channel sum / sum of squares of this tile's fp32 results, accumulated into
  // stats_rows partial rows (tfpp_bn_reduce_final adds the rows in double).  Saves the separate read pass of tfpp_bn_stats.
  if (p.stats_partial) {
    __shared__ float st[2][C::WAVES_M][C::WAVES_N][C::FN][16];
#pragma unroll
    for (int j = 0; j < C::FN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < C::FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) { const float v = acc[i][j][r] * p.alpha; s += v; q += v * v; }
        }
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) { st[0][wm][wn][j][lane] = s; st[1][wm][wn][j][lane] = q; }
    }
    __syncthreads();
    if (wm == 0 && lane < 16) {
      const int ctot = p.G * p.n_g;
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        const int n = bn0 + wn * WN + j * 16 + lane;
        if (n < p.n_g) {
          float s = 0.f, q = 0.f;
          for (int w2 = 0; w2 < C::WAVES_M; ++w2) { s += st[0][w2][wn][j][lane]; q += st[1][w2][wn][j][lane]; }
          // M-tiles are folded onto stats_rows accumulation rows (<= ~100 fp32 atomics per address, pre-zeroed by the caller)
          float* row = p.stats_partial + (size_t)(mtile % p.stats_rows) * 2 * ctot;
          if (p.stats_store) { row[g * p.n_g + n] = s; row[ctot + g * p.n_g + n] = q; }  // one writer per cell: nothing to zero (tfpp.h)
          else { atomicAdd(row + g * p.n_g + n, s); atomicAdd(row + ctot + g * p.n_g + n, q); }
        }
      }
    }
  }

  // epilogue
  if constexpr (sizeof(T) == 2) {
    if (epi_vec_ok(p)) {  // coalesced: 16-row passes through a per-wave LDS strip (see gemm_core.h)
      __syncthreads();
      float* strip = reinterpret_cast<float*>(smem_raw) + wave * EpiStrip<C::FN>::FLOATS;
      if constexpr (BNS) {  // fused BatchNorm-backward statistics of the tensor whose gradient this launch completes (tfpp.h); own instantiation
        BnsAcc<C::FN, C::FM> bns;
        bns.init(p, lane, bn0 + wn * WN, g);
#pragma unroll
        for (int i = 0; i < C::FM; ++i) bns.prefetch(p, lane, i, bm0 + wm * WM + i * 16, M - (bm0 + wm * WM + i * 16), bn0 + wn * WN, g);
#pragma unroll
        for (int i = 0; i < C::FM; ++i) {
          const int m_pass = bm0 + wm * WM + i * 16;
          epi_pass_bf16<C::FN, C::FM, true>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, g, &bns, i);
        }
        __syncthreads();  // the strips are dead
        bns.template finish<C::WAVES_M, C::WAVES_N>(p, reinterpret_cast<float*>(smem_raw), wm, wn, lane, mtile, bn0 + wn * WN, g);
      } else {
#pragma unroll
        for (int i = 0; i < C::FM; ++i) {
          const int m_pass = bm0 + wm * WM + i * 16;
          epi_pass_bf16<C::FN, C::FM, false>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, g);
        }
      }
      return;
    }
  }
  const int hw = p.Hd * p.Wd;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int i = 0; i < C::FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        const int n = bn0 + wn * WN + j * 16 + (lane & 15);
        if (n >= p.n_g) continue;
        const int ch = g * p.n_g + n;
        float v = acc[i][j][r] * p.alpha;
        if (p.scale) v *= p.scale[ch];
        if (p.shift) v += p.shift[ch];
        if (res) v += ElemTraits<T>::to_f(res[(size_t)m * p.res_ld + ch]);
        v = apply_act(v, p.act);
        size_t o;
        if (p.dst_nchw) { const int b = m / hw, pix = m - b * hw; o = ((size_t)b * p.Cd + ch) * hw + pix; }
        else o = (size_t)m * p.dst_ld + ch;
        if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
        else reinterpret_cast<T*>(p.dst)[o] = ElemTraits<T>::from_f(v);
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_conv(const tfpp_conv_params& p, hipStream_t st) {
  // bf16: 64-deep K stages (half the barriers per FLOP; the kernels are latency-bound); fp32 keeps 32 (LDS budget)
  constexpr int BKT = sizeof(T) == 2 ? 64 : 32;
  using C = TileCfg<T, BM, BN, WM, WN, BKT>;
  const long M = (long)p.B * p.Hd * p.Wd;
  dim3 grid(cdiv(p.n_g, BN), cdiv(M, BM), p.G * (p.splitk > 1 ? p.splitk : 1));
  const int K = p.R * p.S * p.ks_g;
  if (BKT == 64 && K <= 32) {  // tiny-K layers (stem): one 32-deep stage is enough
    hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WM, WN, 32>), grid, dim3(C::NT), 0, st, p);
    TFPP_CHECK_LAUNCH();
    return 0;
  }
  if constexpr (sizeof(T) == 2) {
    if (p.bns_partial) {
      hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WM, WN, BKT, true>), grid, dim3(C::NT), 0, st, p);
      TFPP_CHECK_LAUNCH();
      return 0;
    }
  }
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, WM, WN, BKT>), grid, dim3(C::NT), 0, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// tile choice: minimise padded N work, prefer wide tiles; small problems get 64x64 tiles for more workgroups.
// 0: 128x32   1: 128x64   2: 64x64   3: 128x128   4: 128x96 (8 waves of 32x48)
static const int kConvBm[5] = {128, 128, 64, 128, 128}, kConvBn[5] = {32, 64, 64, 128, 96};
static int conv_variant(const tfpp_conv_params& p) {
  const long M = (long)p.B * p.Hd * p.Wd;
  const int N = p.n_g;
  // 72 channels (RegNet stage 1) in ONE 96-wide tile: with 3 x 32 the activation tile is fetched by three workgroups and the launch has
  // three times the workgroups for the same bytes (HBM-bound layers: K = 32..216); bf16 only (fp32 LDS budget), TFPP_CONV_96=0: 3 x 32
  static const int use96 = [] { const char* e = std::getenv("TFPP_CONV_96"); return (e && e[0] == '0') ? 0 : 1; }();
  // (not with the BatchNorm-backward statistics epilogue: its cross-lane reduction needs a power-of-two number of chunks per row)
  if (N > 64 && N <= 96 && use96 && M >= 4096 && !p.bns_partial) return 4;
  if (N <= 32 || (N > 64 && N <= 96)) return 0;  // 24/32 -> one tile, 72 -> 3 x 32
  if (N <= 64) return 1;
  const long tiles128 = (long)cdiv(M, 128) * cdiv(N, 128) * p.G;
  return tiles128 < 1024 ? 2 : 3;  // latency-bound regime: keep >= 4 workgroups per CU in flight
}

static int conv_variant_for(const tfpp_conv_params& p, int dtype) {
  const int v = conv_variant(p);
  return (v == 4 && dtype != TFPP_BF16) ? 0 : v;
}

// Second stage of split-K: dst = epilogue(sum_s ws[s][m][ch]); 4 channels per thread.
template <typename T> __global__ void splitk_epilogue_kernel(tfpp_conv_params p) {
  const int M = p.B * p.Hd * p.Wd, ntot = p.G * p.n_g, nq = ntot >> 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)M * nq) return;
  const int m = (int)(i / nq), ch0 = (int)(i - (long)m * nq) * 4;
  const float* __restrict__ wsp = p.splitk_ws + (size_t)m * ntot + ch0;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sidx = 0; sidx < p.splitk; ++sidx) {
    const float4 v = *reinterpret_cast<const float4*>(wsp + (size_t)sidx * M * ntot);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const float acc[4] = {a.x, a.y, a.z, a.w};
  const int hw = p.Hd * p.Wd;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ch = ch0 + e;
    float v = acc[e] * p.alpha;
    if (p.scale) v *= p.scale[ch];
    if (p.shift) v += p.shift[ch];
    if (res) v += ElemTraits<T>::to_f(res[(size_t)m * p.res_ld + ch]);
    v = apply_act(v, p.act);
    size_t o;
    if (p.dst_nchw) { const int b = m / hw, pix = m - b * hw; o = ((size_t)b * p.Cd + ch) * hw + pix; }
    else o = (size_t)m * p.dst_ld + ch;
    if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
    else reinterpret_cast<T*>(p.dst)[o] = ElemTraits<T>::from_f(v);
  }
}

template <typename T> static int launch_splitk_epilogue(const tfpp_conv_params& p, hipStream_t st) {
  const long n = (long)p.B * p.Hd * p.Wd * ((p.G * p.n_g) >> 2);
  hipLaunchKernelGGL(splitk_epilogue_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// K slices for a problem that yields `tiles` output tiles with K stages of depth bk: only when the grid would leave most
// of the chip idle (<= 192 workgroups) and every slice keeps >= TFPP_SPLITK_MIN_STAGES K stages (default 2).  Round 5 measured 12 stages per
// slice (140 fewer splitk_epilogue launches of the 792 in the bs = 1 forward): 3.95 -> 4.37 ms bf16, 5.95 -> 7.9 ms fp32, training step
// +0.6 ms -- the second launch costs less than the serial K loop of a few workgroups (profiles/r05_ab_se_splitk.txt).
static int conv_splits(const tfpp_conv_params& p, long tiles, int bk) {
  static const int min_stages = [] { const char* e = std::getenv("TFPP_SPLITK_MIN_STAGES"); const int v = e ? std::atoi(e) : 2; return v < 2 ? 2 : v; }();
  if (!p.splitk_ws || p.stats_partial || p.bns_partial || tiles > 192) return 1;
  const int K = p.R * p.S * p.ks_g, ntot = p.G * p.n_g;
  if (ntot % 4) return 1;
  long sp = (512 + tiles - 1) / tiles;
  if (sp > K / (min_stages * bk)) sp = K / (min_stages * bk);
  if (sp > 32) sp = 32;
  const long per_slice = (long)p.B * p.Hd * p.Wd * ntot;
  while (sp > 1 && sp * per_slice > p.splitk_ws_floats) --sp;
  return sp < 2 ? 1 : (int)sp;
}

// number of M-tiles (= rows of stats_partial the LDS kernel writes) for p
extern "C" int tfpp_conv_gemm_mtiles(const tfpp_conv_params* p) {
  if (!p) return TFPP_EINVAL;
  if (conv_halo_supported(*p, TFPP_BF16)) {  // dtype unknown here: an upper bound over both kernels is enough
    const int a = conv_halo_mtiles(*p), b = cdiv((long)p->B * p->Hd * p->Wd, 64);
    return a > b ? a : b;
  }
  return cdiv((long)p->B * p->Hd * p->Wd, kConvBm[conv_variant(*p)]);  // (variant 4 and 0 share BM) only an upper bound for the row count is needed
}

// TFPP_CONV_GLDS=0 disables the multi-stage LDS-DMA kernel (gemm_glds.hip) for A/B measurements
static bool use_glds_impl() {
  static const int v = [] {
    const char* e = std::getenv("TFPP_CONV_GLDS");
    return (e && std::strcmp(e, "0") == 0) ? 0 : 1;
  }();
  return v != 0;
}

extern "C" int tfpp_conv_gemm_variant(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  if (conv_halo_supported(*p, dtype)) return conv_halo_variant(*p);
  if (use_glds_impl() && conv_glds_supported(*p, dtype)) return conv_glds_variant(*p);
  return conv_variant_for(*p, dtype);
}

static int conv_splits_for(const tfpp_conv_params& p, int dtype) {
  const long M = (long)p.B * p.Hd * p.Wd;
  if (conv_halo_supported(p, dtype)) return 1;
  if (use_glds_impl() && conv_glds_supported(p, dtype)) {
    const int var = conv_glds_variant(p), bm = conv_glds_bm(var);
    if (var == 202) return 1;  // >= 128 workgroups of 16 waves with >= 16 stages each: splitting K only adds the second pass
    return conv_splits(p, (long)cdiv(M, bm) * cdiv(p.n_g, 128) * p.G, 64);
  }
  const int v = conv_variant_for(p, dtype);
  return conv_splits(p, (long)cdiv(M, kConvBm[v]) * cdiv(p.n_g, kConvBn[v]) * p.G, dtype == TFPP_BF16 ? 64 : 32);
}

extern "C" int tfpp_conv_gemm_splits(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  return conv_splits_for(*p, dtype);
}

// fused BatchNorm-backward statistics need the vector epilogue (bf16 NHWC destination, 8-channel granularity, aligned bases)
static bool conv_bns_ok(const tfpp_conv_params& p, int dtype) {
  if (dtype != TFPP_BF16 || p.dst_nchw || p.dst_f32 || ((p.n_g | (int)p.dst_ld | (int)p.bns_ld) & 7) || ((uintptr_t)p.dst & 15)) return false;
  if (p.res && ((((int)p.res_ld) & 7) || ((uintptr_t)p.res & 15))) return false;
  if (p.R * p.S * p.ks_g <= 32) return false;  // the one-stage tiny-K instantiation has no statistics variant
  return true;
}
extern "C" int tfpp_conv_gemm_bns_ok(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  tfpp_conv_params q = *p;
  q.bns_partial = reinterpret_cast<float*>(16);  // plan as if the feature were on (it disables split-K)
  if (q.bns_ld == 0) q.bns_ld = q.dst_ld;
  return conv_bns_ok(q, dtype) ? 1 : 0;
}

// ReLU backward in the epilogue (tfpp_conv_params.relu_mask): the vector epilogue of a launch without a K split
static bool conv_relu_mask_ok(const tfpp_conv_params& p, int dtype) {
  if (dtype != TFPP_BF16 || p.dst_nchw || p.dst_f32 || ((p.n_g | (int)p.dst_ld | (int)p.relu_mask_ld) & 7) || ((uintptr_t)p.dst & 15)) return false;
  if (((uintptr_t)p.relu_mask & 15) || p.bns_partial) return false;
  if (p.res && ((((int)p.res_ld) & 7) || ((uintptr_t)p.res & 15))) return false;
  if (p.R * p.S * p.ks_g <= 32) return false;
  return conv_splits_for(p, dtype) == 1;
}
extern "C" int tfpp_conv_gemm_relu_mask_ok(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  tfpp_conv_params q = *p;
  if (!q.relu_mask) q.relu_mask = reinterpret_cast<const void*>(16);
  if (q.relu_mask_ld == 0) q.relu_mask_ld = q.dst_ld;
  return conv_relu_mask_ok(q, dtype) ? 1 : 0;
}

extern "C" int tfpp_conv_gemm_in_bn_ok(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  return conv_halo_in_bn_ok(*p, dtype) ? 1 : 0;
}

// exact number of M-tiles (= distinct stats_partial rows) of the kernel the dispatcher runs for (p, dtype)
extern "C" int tfpp_conv_gemm_stats_rows(const tfpp_conv_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  const long M = (long)p->B * p->Hd * p->Wd;
  if (conv_halo_supported(*p, dtype)) return conv_halo_mtiles(*p);
  if (use_glds_impl() && conv_glds_supported(*p, dtype)) return cdiv(M, conv_glds_bm(conv_glds_variant(*p)));
  return cdiv(M, kConvBm[conv_variant_for(*p, dtype)]);
}

template <typename T> static int dispatch_conv(const tfpp_conv_params& p, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (p.ks_g % VEC != 0 || p.src_ld % VEC != 0 || p.G < 1 || p.B < 1) return TFPP_EINVAL;
  if (((uintptr_t)p.src & 15) || ((uintptr_t)p.w & 15)) return TFPP_EINVAL;
  const long M = (long)p.B * p.Hd * p.Wd;
  if (M >= (1l << 31) || M * (long)p.dst_ld >= (1l << 40)) return TFPP_EINVAL;
  if (p.bns_partial && (!conv_bns_ok(p, ElemTraits<T>::DT) || !p.bns_x || !p.bns_mean || !p.bns_invstd || (p.bns_relu && !p.bns_y)))
    return TFPP_EINVAL;  // the caller asks tfpp_conv_gemm_bns_ok first
  if (p.in_bn.scale && !conv_halo_in_bn_ok(p, ElemTraits<T>::DT)) return TFPP_EINVAL;  // the caller asks tfpp_conv_gemm_in_bn_ok first
  if (p.relu_mask && !conv_relu_mask_ok(p, ElemTraits<T>::DT)) return TFPP_EINVAL;      // the caller asks tfpp_conv_gemm_relu_mask_ok first
  if (p.stats_partial && p.stats_store && p.stats_rows != tfpp_conv_gemm_stats_rows(&p, ElemTraits<T>::DT)) return TFPP_EINVAL;
  if (conv_halo_supported(p, ElemTraits<T>::DT)) return conv_gemm_halo(p, st);
  tfpp_conv_params q = p;
  q.splitk = conv_splits_for(p, ElemTraits<T>::DT);
  int rc;
  if (use_glds_impl() && conv_glds_supported(p, ElemTraits<T>::DT)) rc = conv_gemm_glds(q, st);
  else {
    switch (conv_variant_for(p, ElemTraits<T>::DT)) {
      case 4:
        if constexpr (sizeof(T) == 2) {
          rc = launch_conv<T, 128, 96, 32, 48>(q, st);  // (64x96 tiles measured the same: 24.1 vs 24.7 us on 196608x72x72)
          break;
        }
      case 0: rc = launch_conv<T, 128, 32, 32, 32>(q, st); break;
      case 1: rc = launch_conv<T, 128, 64, 64, 32>(q, st); break;
      case 2: rc = launch_conv<T, 64, 64, 32, 32>(q, st); break;
      default: rc = launch_conv<T, 128, 128, 64, 64>(q, st); break;
    }
  }
  if (rc != 0 || q.splitk <= 1) return rc;
  return launch_splitk_epilogue<T>(q, st);
}

extern "C" int tfpp_conv_gemm(const tfpp_conv_params* p, int dtype, void* stream) {
  if (!p || !p->src || !p->w || !p->dst) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) return dispatch_conv<float>(*p, st);
  if (dtype == TFPP_BF16) return dispatch_conv<bf16_t>(*p, st);
  return TFPP_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[n][(c,r,s)] += sum_pixels dY[pix][n] * Xgather[pix][(r,s,c)]   (reduction over pixels)
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int BKT>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * TFPP_WAVE) void conv_wgrad_kernel(tfpp_wgrad_params p) {
  using C = TileCfg<T, BM, BN, WM, WN, BKT>;
  constexpr int VEC = C::VEC, NT = C::NT, BK = C::BK;
  __shared__ __attribute__((aligned(16))) T As[C::A_ELEMS_KM];
  __shared__ __attribute__((aligned(16))) T Bs[C::B_ELEMS_KM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  // grid: x = (group, pixel split) with the split fastest, y = output-row tile, z = column tile.  All tiles of one pixel slab
  // then run on XCD (split % 8): the slab of dY / X is fetched into that L2 once and shared by every (n, kk) tile.
  const int g = blockIdx.x / p.splits, split = blockIdx.x - g * p.splits;
  const int bm0 = blockIdx.y * BM, bn0 = blockIdx.z * BN;
  const int KK = p.R * p.S * p.ks_g;
  const long P = (long)p.B * p.Hd * p.Wd;
  const long per = ((P + p.splits - 1) / p.splits + BK - 1) / BK * BK;
  const long p_beg = (long)split * per, p_end = (p_beg + per < P) ? p_beg + per : P;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy);
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x);

  constexpr int AV = BM / VEC, BV = BN / VEC;           // vectors per pixel row of each tile
  constexpr int A_IT = (BK * AV + NT - 1) / NT, B_IT = (BK * BV + NT - 1) / NT;
  f32x4_t acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int hw = p.Hd * p.Wd;
  // register-prefetched pixel loop (round 6, as conv_gemm_kernel): the loads of stage pt + BK are in flight during the MFMAs of stage pt
  uint4 a_val[A_IT], b_val[B_IT];
  auto load_stage = [&](long pt) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (v < BK * AV) {
        const int pk = v / AV, nc = v - pk * AV, n = bm0 + nc * VEC;
        const long pix = pt + pk;
        if (pix < p_end && n < p.n_g) val = *reinterpret_cast<const uint4*>(dy + (size_t)pix * p.dy_ld + g * p.n_g + n);
      }
      a_val[i] = val;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * NT;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (v < BK * BV) {
        const int pk = v / BV, kc = v - pk * BV, kk0 = bn0 + kc * VEC;
        const long pix = pt + pk;
        if (pix < p_end && kk0 < KK) {
          const int rs = kk0 / p.ks_g, c = kk0 - rs * p.ks_g, r = rs / p.S, s = rs - r * p.S;
          const int b = (int)(pix / hw), rem = (int)(pix - (long)b * hw), hd = rem / p.Wd, wd = rem - hd * p.Wd;
          const int hs = hd * p.stride - p.pad + r, ws = wd * p.stride - p.pad + s;
          if (hs >= 0 && hs < p.Hs && ws >= 0 && ws < p.Ws)
            val = *reinterpret_cast<const uint4*>(x + ((size_t)(b * p.Hs + hs) * p.Ws + ws) * p.x_ld + g * p.ks_g + c);
        }
      }
      b_val[i] = val;
    }
  };
  if (p_beg < p_end) load_stage(p_beg);
  for (long pt = p_beg; pt < p_end; pt += BK) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int v = tid + i * NT;
      if (v < BK * AV) {
        const int pk = v / AV, nc = v - pk * AV;
        lds_store_km<T>(&As[pk * C::LDRA + nc * VEC], a_val[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int v = tid + i * NT;
      if (v < BK * BV) {
        const int pk = v / BV, kc = v - pk * BV;
        lds_store_km<T>(&Bs[pk * C::LDRB + kc * VEC], b_val[i]);
      }
    }
    __syncthreads();
    if (pt + BK < p_end) load_stage(pt + BK);
    tile_mma_step<C, T, true, true>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }

  const int RS = p.R * p.S;
  if (p.ws && p.splits > 1) {  // slice -> workspace [split][G*n_g][KK]; wgrad_reduce_kernel sums the slices into dw
    float* __restrict__ wsp = p.ws + ((size_t)split * p.G * p.n_g + (size_t)g * p.n_g) * KK;
#pragma unroll
    for (int i = 0; i < C::FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
        if (n >= p.n_g) continue;
#pragma unroll
        for (int j = 0; j < C::FN; ++j) {
          const int kk = bn0 + wn * WN + j * 16 + (lane & 15);
          if (kk < KK) wsp[(size_t)n * KK + kk] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < C::FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
      if (n >= p.n_g) continue;
      int row = g * p.n_g + n;
      if (p.row_map) row = p.row_map[row];
      if (row < 0) continue;
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        const int kk = bn0 + wn * WN + j * 16 + (lane & 15);
        if (kk >= KK) continue;
        long col;
        if (p.col_map) {
          col = p.col_map[kk];
          if (col < 0) continue;
        } else {
          const int rs = kk / p.ks_g, c = kk - rs * p.ks_g;
          if (c >= p.c_real) continue;
          col = (long)c * RS + rs;
        }
        float* o = p.dw + (size_t)row * p.dw_ld + col;
        if (p.splits > 1) atomicAdd(o, acc[i][j][r]);
        else *o += acc[i][j][r];  // single writer
      }
    }
  }
}

// dw[row][col] += sum_s ws[s][row][kk]  (second stage of the pixel-split weight gradient).  A workgroup is SLOTS slice
// slots x (256 / SLOTS) consecutive (row, kk) elements: many slices of a small gradient (stage-1 layers: 384 slices of
// 72 x 72) are summed by 16 slots in parallel, few slices of a large one by 4.
template <int SLOTS> __device__ __forceinline__ void wgrad_reduce_body(const tfpp_wgrad_params& p, const long block) {
  constexpr int EL = 256 / SLOTS;
  const int KK = p.R * p.S * p.ks_g, rows = p.G * p.n_g;
  const int el = threadIdx.x % EL, slot = threadIdx.x / EL;
  const long i = block * EL + el;
  const size_t slice = (size_t)rows * KK;
  float s = 0.f;
  if (i < (long)slice) {
    const float* __restrict__ wsp = p.ws + i;
#pragma unroll 4
    for (int k = slot; k < p.splits; k += SLOTS) s += wsp[(size_t)k * slice];
  }
  __shared__ float sm[SLOTS][EL + 1];
  sm[slot][el] = s;
  __syncthreads();
  if (slot != 0 || i >= (long)slice) return;
#pragma unroll
  for (int k = 1; k < SLOTS; ++k) s += sm[k][el];
  const int r = (int)(i / KK), kk = (int)(i - (long)r * KK);
  int row = r;
  if (p.row_map) row = p.row_map[r];
  if (row < 0) return;
  long col;
  if (p.col_map) {
    col = p.col_map[kk];
    if (col < 0) return;
  } else {
    const int rs = kk / p.ks_g, c = kk - rs * p.ks_g;
    if (c >= p.c_real) return;
    col = (long)c * (p.R * p.S) + rs;
  }
  p.dw[(size_t)row * p.dw_ld + col] += s;
}
template <int SLOTS> __global__ void wgrad_reduce_kernel(tfpp_wgrad_params p) { wgrad_reduce_body<SLOTS>(p, (long)blockIdx.x); }

// slice sums of every pixel-split layer of a grouped weight-gradient launch in ONE grid (wg_start / wgs of the items now count the
// workgroups of this kernel: ceil(n_g * KK / 64) per layer)
__global__ void wgrad_reduce_group_kernel(const tfpp_wgrad_group grp) {
  const int id = (int)blockIdx.x;
  int k = 0;
  for (int i = 1; i < grp.n; ++i) k = (id >= grp.it[i].wg_start) ? i : k;
  k = __builtin_amdgcn_readfirstlane(k);
  const int local = id - grp.it[k].wg_start;
  if (local >= grp.it[k].wgs) return;
  const tfpp_wgrad_params p = tfpp_wgrad_item_params(grp.it[k]);
  wgrad_reduce_body<4>(p, (long)local);
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_wgrad(const tfpp_wgrad_params& p, hipStream_t st) {
  constexpr int BKT = sizeof(T) == 2 ? 64 : 32;
  using C = TileCfg<T, BM, BN, WM, WN, BKT>;
  const int KK = p.R * p.S * p.ks_g;
  dim3 grid(p.G * p.splits, cdiv(p.n_g, BM), cdiv(KK, BN));
  hipLaunchKernelGGL((conv_wgrad_kernel<T, BM, BN, WM, WN, BKT>), grid, dim3(C::NT), 0, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// The plan of one weight-gradient call: which kernel, how many pixel slices, whether a second-stage sum follows.
// variant: 0 = LDS-staged 32x32, 1 = LDS-staged 64x64, 2 = LDS-DMA ring 64x64, 3 = 3x3 halo (wgrad3x3_halo.hip),
//          4 = LDS-DMA ring 128x128 with 8 waves (gemm_wgrad_glds.hip)
struct WgradPlan { int variant, splits, reduce; };
template <typename T> static int plan_wgrad(tfpp_wgrad_params& p, WgradPlan& pl) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (p.ks_g % VEC != 0 || p.n_g % VEC != 0 || p.x_ld % VEC != 0 || p.dy_ld % VEC != 0) return TFPP_EINVAL;
  const long P = (long)p.B * p.Hd * p.Wd;
  const int KK = p.R * p.S * p.ks_g;
  if (const int hs = wgrad_halo_slices(p, ElemTraits<T>::DT)) {  // 3x3 stride 1, few channels: halo tiles staged once in LDS
    p.splits = hs;
    pl = WgradPlan{3, hs, 1};
    return 0;
  }
  const bool small = (p.n_g <= 32 || KK <= 32);
  const bool glds = !small && wgrad_glds_supported(p, ElemTraits<T>::DT);
  // wide layers with enough work to amortise the deeper ring: 128x128 tiles, 8 waves (>= 4 GFLOP per group keeps >= ~8 stages per
  // workgroup at one round of the chip)
  const bool big = glds && wgrad_glds128_preferred(p) && 2.0 * (double)P * p.n_g * KK >= 4e9;
  const int bm = small ? 32 : (big ? 128 : 64), bn = bm;
  const long tiles = (long)cdiv(p.n_g, bm) * cdiv(KK, bn) * p.G;
  if (p.splits <= 0) {
    long want, maxs = (P + 511) / 512;  // at least 512 pixels of reduction per workgroup
    if (big) want = 256 / tiles;        // one workgroup (96 KB of LDS) per CU: about one round of the chip, no pixel split from 256 tiles
    else want = (1536 + tiles - 1) / tiles;  // 4-wave workgroups: enough of them to fill 256 CUs several times over
    p.splits = (int)(want < 1 ? 1 : (want > maxs ? maxs : want));
    if (p.splits >= 8) p.splits = p.splits / 8 * 8;  // whole XCD rounds
    if (p.splits < 1) p.splits = 1;
  }
  const long slice = (long)p.G * p.n_g * KK;
  if (p.ws && p.splits > 1 && (long)p.splits * slice > p.ws_floats) {  // shrink to what the workspace holds
    const long fit = p.ws_floats / slice;
    if (fit >= 2) p.splits = (int)fit;
    else p.ws = nullptr;  // atomics
  }
  pl.variant = small ? 0 : (glds ? (big ? 4 : 2) : 1);
  pl.splits = p.splits;
  pl.reduce = (p.ws && p.splits > 1) ? 1 : 0;
  return 0;
}

template <typename T> static int run_wgrad_stage1(const tfpp_wgrad_params& p, const WgradPlan& pl, hipStream_t st) {
  switch (pl.variant) {
    case 4: return conv_wgrad_glds(p, 128, st);
    case 3: return conv_wgrad_halo(p, pl.splits, st);
    case 2: return conv_wgrad_glds(p, 64, st);
    case 1: return launch_wgrad<T, 64, 64, 32, 32>(p, st);
    default: return launch_wgrad<T, 32, 32, 16, 16>(p, st);
  }
}

static int run_wgrad_reduce(const tfpp_wgrad_params& p, hipStream_t st) {
  const long slice = (long)p.G * p.n_g * p.R * p.S * p.ks_g;
  if (p.splits >= 32) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3((unsigned)((slice + 15) / 16)), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((slice + 63) / 64)), dim3(256), 0, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// stage: 0 = both kernels, 1 = first stage only, 2 = slice sum only.  plan_out (nullable): {variant, splits, reduce}.
template <typename T> static int dispatch_wgrad(tfpp_wgrad_params p, int stage, int* plan_out, hipStream_t st) {
  WgradPlan pl;
  int rc = plan_wgrad<T>(p, pl);
  if (rc != 0) return rc;
  if (plan_out) { plan_out[0] = pl.variant; plan_out[1] = pl.splits; plan_out[2] = pl.reduce; }
  if (p.x_scale && (pl.variant != 3 || !p.x_shift)) return TFPP_EINVAL;  // normalise-on-load: 3x3 halo kernel only (tfpp_conv_wgrad_x_bn_ok)
  if (stage < 0) return 0;  // plan only
  if (stage != 2) rc = run_wgrad_stage1<T>(p, pl, st);
  if (rc != 0 || stage == 1 || !pl.reduce) return rc;
  return run_wgrad_reduce(p, st);
}

extern "C" int tfpp_conv_wgrad_x_bn_ok(const tfpp_wgrad_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  return (dtype == TFPP_BF16 && wgrad_halo_slices(*p, dtype) > 0) ? 1 : 0;
}

extern "C" int tfpp_conv_wgrad(const tfpp_wgrad_params* p, int dtype, void* stream) {
  if (!p || !p->dy || !p->x || !p->dw) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) return dispatch_wgrad<float>(*p, 0, nullptr, st);
  if (dtype == TFPP_BF16) return dispatch_wgrad<bf16_t>(*p, 0, nullptr, st);
  return TFPP_EINVAL;
}

// ---- many weight gradients in one call (round 5) --------------------------------------------------------------------------------------
// The weight-gradient lane of the training step hands over a whole batch of independent layers (engine.SideLane.flush).  Pointwise bf16
// layers the LDS-DMA kernel covers are launched as grouped grids (conv_wgrad_glds_group_kernel: one grid per <= 42 layers of a tile
// class, descriptor table in the kernel arguments); everything else runs through the single-layer dispatcher, in the order given.
// Pixel splits are chosen for the GROUP: the grid as a whole has to fill the chip, not every layer on its own -- the stage-3 / stage-4 /
// fusion-transformer layers then need no split at all (their tile is added straight into dW: no slices, no second pass), layers with
// few tiles and many pixels (stages 1-2) keep slices + ONE grouped slice-sum launch (deterministic order, no atomics).
// Two layers of a batch that write the same dW (a parameter used twice) never share a group: the later one takes the single-layer path.
namespace {
struct GroupBuild {
  int tile, bkp;
  std::vector<tfpp_wgrad_params> items;
};
}  // namespace

static int emit_wgrad_group(const std::vector<tfpp_wgrad_params>& its, int tile, int bkp, int grid_cap, hipStream_t st) {
  // target: ~4 workgroups per CU-slot over the whole group, every workgroup >= 512 pixels of reduction
  static const int target = [] { const char* e = std::getenv("TFPP_WGRAD_GROUP_TARGET"); const int v = e ? std::atoi(e) : 0; return v; }();
  // (swept on the bs = 12 step, profiles/r05_ab_wgrad_group_target.txt: 256: +0.4 ms, 512: +0.05, 2048: -0.03, 6144: +0.1 against the ungrouped lane)
  const long want_wgs = target > 0 ? target : 2048;
  static const int pin_on = [] { const char* e = std::getenv("TFPP_WGRAD_PIN"); return (e && e[0] == '0') ? 0 : 1; }();
  static const long pin_max_tiles = [] { const char* e = std::getenv("TFPP_WGRAD_PIN_MAX_TILES"); return e ? std::atol(e) : 48l; }();
  static const double pin_slab_bytes = [] { const char* e = std::getenv("TFPP_WGRAD_PIN_SLAB_KB"); return (e ? std::atof(e) : 3072.0) * 1024.0; }();
  size_t pos = 0;
  while (pos < its.size()) {
    const size_t n = std::min(its.size() - pos, (size_t)TFPP_WGRAD_GROUP_MAX);
    // work of this chunk in (tile x stage) units
    double work = 0;
    for (size_t i = 0; i < n; ++i) {
      const tfpp_wgrad_params& q = its[pos + i];
      const long P = (long)q.B * q.Hd * q.Wd;
      work += (double)cdiv(q.n_g, tile) * cdiv(q.ks_g, tile) * (double)cdiv(P, bkp);
    }
    const double min_grain = 512.0 / bkp;
    double grain = work / (double)want_wgs;
    if (grain < min_grain) grain = min_grain;
    tfpp_wgrad_group grp, red;
    grp.n = red.n = 0;
    grp.total = red.total = 0;
    float* ws = its[pos].ws;
    long ws_left = ws ? its[pos].ws_floats : 0, ws_off = 0;
    for (size_t i = 0; i < n; ++i) {
      const tfpp_wgrad_params& q = its[pos + i];
      const long P = (long)q.B * q.Hd * q.Wd, stages = cdiv(P, bkp), slice = (long)q.n_g * q.ks_g;
      long sp = (long)(stages / grain + 0.5);
      if (sp < 1) sp = 1;
      if (sp > 64) sp = 64;
      if (sp >= 8) sp = sp / 8 * 8;
      // Round 6: a layer with a small tile grid (stage-2 / stage-3 convolutions: 16-25 tiles, thousands of pixels) cannot give every XCD a
      // block of tiles that share operand panels -- dealt round-robin over the XCDs every tile streamed its two panels from the fabric
      // (157 MB for a 576 x 576 x 12288 layer whose operands are 28 MB).  Its pixel slices become units PINNED to one XCD each, sized so
      // that the (tiles_m + tiles_n) panels of a slice fit that XCD's L2 (TFPP_WGRAD_PIN=0: the round-5 orders).
      const long tm = cdiv(q.n_g, tile), tn = cdiv(q.ks_g, tile);
      int pin = -1;
      if (pin_on && tile <= 128 && tm * tn <= pin_max_tiles) {
        const double slab = (double)(tm + tn) * tile * 2.0 * (double)P;  // bytes of all panels over all pixels
        long fit = (long)(slab / pin_slab_bytes + 0.999);
        const long max_sp = stages / 4 > 0 ? stages / 4 : 1;  // >= 4 stages per slice
        if (fit > max_sp) fit = max_sp;
        if (sp < fit) sp = fit;
        if (sp > 64) sp = 64;
        if (sp >= 8) sp = (sp + 7) / 8 * 8 <= 64 ? (sp + 7) / 8 * 8 : 64;
        pin = (int)(i & 7);
      }
      if (sp > 1 && sp * slice > ws_left - ws_off) sp = (ws_left - ws_off) / slice >= 2 ? (ws_left - ws_off) / slice : 1;
      if ((sp & 7) == 0) pin = -1;  // whole XCD rounds of slices: the kernel's slice-per-XCD order already does this
      tfpp_wgrad_item& it = grp.it[grp.n++];
      it.dy = q.dy; it.x = q.x; it.dw = q.dw; it.row_map = q.row_map; it.col_map = q.col_map;
      it.P = (int)P; it.n_g = q.n_g; it.KK = q.ks_g; it.c_real = q.c_real; it.splits = (int)sp;
      it.dy_ld = (int)q.dy_ld; it.x_ld = (int)q.x_ld; it.dw_ld = (int)q.dw_ld;
      it.ws = sp > 1 ? ws + ws_off : nullptr;
      it.wg_start = grp.total;
      it.pin = pin; it.pad_ = 0;
      it.wgs = (int)(pin >= 0 ? (sp + 7) / 8 * 8 : sp) * cdiv(q.n_g, tile) * cdiv(q.ks_g, tile);
      grp.total += (it.wgs + 7) / 8 * 8;
      if (sp > 1) {
        ws_off += sp * slice;
        tfpp_wgrad_item& r = red.it[red.n++];
        r = it;
        r.wg_start = red.total;
        r.wgs = (int)((slice + 63) / 64);
        red.total += r.wgs;
      }
    }
    int rc = conv_wgrad_glds_group(grp, tile, grid_cap, st);
    if (rc != 0) return rc;
    if (red.n > 0) {
      hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)red.total), dim3(256), 0, st, red);
      TFPP_CHECK_LAUNCH();
    }
    pos += n;
  }
  return 0;
}

// 1 if tfpp_conv_wgrad_batch would put this layer into a grouped grid (bench.py's per-kernel bookkeeping; the tile class is the plan's variant)
extern "C" int tfpp_conv_wgrad_group_ok(const tfpp_wgrad_params* p, int dtype) {
  if (!p) return TFPP_EINVAL;
  return (dtype == TFPP_BF16 && wgrad_glds_group_ok(*p, dtype)) ? 1 : 0;
}

extern "C" int tfpp_conv_wgrad_batch(const tfpp_wgrad_params* items, int n, int dtype, void* stream) {
  if (!items || n < 0 || (dtype != TFPP_F32 && dtype != TFPP_BF16)) return TFPP_EINVAL;
  static const int grid_cap = [] { const char* e = std::getenv("TFPP_WGRAD_GROUP_WGS"); return e ? std::atoi(e) : 0; }();
  hipStream_t st = (hipStream_t)stream;
  GroupBuild g64{64, 32, {}}, g128{128, 64, {}}, g256{256, 32, {}};
  static const int t256 = [] { const char* e = std::getenv("TFPP_WGRAD_GROUP_256"); return (e && e[0] == '0') ? 0 : 1; }();
  std::vector<const float*> seen;
  for (int i = 0; i < n; ++i) {
    tfpp_wgrad_params q = items[i];
    if (!q.dy || !q.x || !q.dw) return TFPP_EINVAL;
    bool grouped = false;
    if (dtype == TFPP_BF16 && wgrad_glds_group_ok(q, dtype) && std::find(seen.begin(), seen.end(), q.dw) == seen.end()) {
      WgradPlan pl;
      tfpp_wgrad_params t = q;
      if (plan_wgrad<bf16_t>(t, pl) == 0 && (pl.variant == 2 || pl.variant == 4)) {
        const bool wide = t256 && pl.variant == 4 && q.n_g >= 1024 && q.ks_g >= 1024;
        (wide ? g256 : (pl.variant == 4 ? g128 : g64)).items.push_back(q);
        grouped = true;
      }
    }
    seen.push_back(q.dw);
    if (grouped) continue;
    const int rc = dtype == TFPP_F32 ? dispatch_wgrad<float>(q, 0, nullptr, st) : dispatch_wgrad<bf16_t>(q, 0, nullptr, st);
    if (rc != 0) return rc;
  }
  for (GroupBuild* g : {&g256, &g128, &g64}) {
    if (g->items.empty()) continue;
    // longest pixel reductions first: the workgroups that run longest are dispatched first
    std::stable_sort(g->items.begin(), g->items.end(), [](const tfpp_wgrad_params& a, const tfpp_wgrad_params& b) {
      return (long)a.B * a.Hd * a.Wd > (long)b.B * b.Hd * b.Wd;
    });
    const int rc = emit_wgrad_group(g->items, g->tile, g->bkp, grid_cap, st);
    if (rc != 0) return rc;
  }
  return 0;
}

// Workspace the dispatcher would like for one call (the library never allocates, the caller passes workspace in):
// the plan is made as if the workspace were unlimited and its size returned; a smaller (or no) workspace only reduces the number of
// K / pixel slices.  op 0: tfpp_conv_gemm split-K slices (tfpp_conv_params.splitk_ws), op 1: tfpp_conv_wgrad pixel slices
// (tfpp_wgrad_params.ws), op 2: BatchNorm scratch for C = *(const int*)params channels, op 3: column-sum scratch for C channels.
extern "C" int tfpp_workspace_bytes(int op, const void* params, int dtype, int64_t* bytes_out) {
  if (!params || !bytes_out) return TFPP_EINVAL;
  *bytes_out = 0;
  if (op == 0) {
    tfpp_conv_params q = *static_cast<const tfpp_conv_params*>(params);
    q.splitk_ws = reinterpret_cast<float*>(16);
    q.splitk_ws_floats = (int64_t)1 << 60;
    const int sp = conv_splits_for(q, dtype);
    if (sp > 1) *bytes_out = (int64_t)sp * q.B * q.Hd * q.Wd * q.G * q.n_g * 4;
    return 0;
  }
  if (op == 1) {
    tfpp_wgrad_params q = *static_cast<const tfpp_wgrad_params*>(params);
    q.ws = reinterpret_cast<float*>(16);
    q.ws_floats = (int64_t)1 << 60;
    WgradPlan pl;
    const int rc = dtype == TFPP_F32 ? plan_wgrad<float>(q, pl) : plan_wgrad<bf16_t>(q, pl);
    if (rc != 0) return rc;
    if (pl.reduce) *bytes_out = (int64_t)pl.splits * q.G * q.n_g * q.R * q.S * q.ks_g * 4;
    return 0;
  }
  if (op == 2 || op == 3) {
    const int C = *static_cast<const int*>(params);
    if (C < 1) return TFPP_EINVAL;
    *bytes_out = (int64_t)(op == 2 ? tfpp_bn_scratch_floats(C) : tfpp_reduce_scratch_floats(1, C)) * 4;
    return 0;
  }
  return TFPP_EINVAL;
}

// The same in separately launchable pieces (per-kernel timing in bench.py): stage 1 = first-stage kernel, 2 = slice sum,
// -1 = plan only.  plan_out[3] = {variant (0 LDS 32x32, 1 LDS 64x64, 2 LDS-DMA ring 64x64, 3 3x3 halo, 4 LDS-DMA ring 128x128), slices, has second stage}.
extern "C" int tfpp_conv_wgrad_stage(const tfpp_wgrad_params* p, int dtype, int stage, int* plan_out, void* stream) {
  if (!p || !p->dy || !p->x || !p->dw || stage == 0 || stage > 2) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) return dispatch_wgrad<float>(*p, stage, plan_out, st);
  if (dtype == TFPP_BF16) return dispatch_wgrad<bf16_t>(*p, stage, plan_out, st);
  return TFPP_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------
// strided batched GEMM (attention products and their gradients); scalar-load fallback for unaligned shapes
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool A_KM, bool B_KM>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * TFPP_WAVE) void bgemm_kernel(tfpp_bgemm_params p, int vec_ok) {
  using C = TileCfg<T, BM, BN, WM, WN>;
  constexpr int VEC = C::VEC, KV = C::KV, NT = C::NT, BK = C::BK;
  __shared__ __attribute__((aligned(16))) T As[A_KM ? C::A_ELEMS_KM : C::A_ELEMS_RM];
  __shared__ __attribute__((aligned(16))) T Bs[B_KM ? C::B_ELEMS_KM : C::B_ELEMS_RM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WAVES_N, wn = wave % C::WAVES_N;
  const int z = blockIdx.z, z0 = z / p.batch1, z1 = z - z0 * p.batch1;
  const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1;
  const T* __restrict__ Bp = reinterpret_cast<const T*>(p.B) + z0 * p.b_bs0 + z1 * p.b_bs1;
  const size_t coff = (size_t)z0 * p.c_bs0 + (size_t)z1 * p.c_bs1;

  f32x4_t acc[C::FM][C::FN];
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + BK - 1) / BK;
  if (vec_ok) {
    // aligned operands (round 6): register-prefetched K loop -- the 16-byte loads of tile kt + 1 are in flight during the MFMAs of tile kt
    constexpr int A_RV = BM / VEC, B_RV = BN / VEC;
    constexpr int AIT = A_KM ? (BK * A_RV + NT - 1) / NT : (BM * KV + NT - 1) / NT;
    constexpr int BIT = B_KM ? (BK * B_RV + NT - 1) / NT : (BN * KV + NT - 1) / NT;
    uint4 av[AIT], bv[BIT];
    auto load_regs = [&](int kt) {
      const int kb = kt * BK;
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        const int v = tid + i * NT;
        uint4 val = make_uint4(0, 0, 0, 0);
        if constexpr (!A_KM) {
          if (v < BM * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, m = bm0 + row;
            if (m < p.M && k0 < p.K) val = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + k0);
          }
        } else {
          if (v < BK * A_RV) {
            const int kk = v / A_RV, rc = v - kk * A_RV, k = kb + kk, m0 = bm0 + rc * VEC;
            if (k < p.K && m0 < p.M) val = *reinterpret_cast<const uint4*>(A + (size_t)k * p.lda + m0);
          }
        }
        av[i] = val;
      }
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int v = tid + i * NT;
        uint4 val = make_uint4(0, 0, 0, 0);
        if constexpr (!B_KM) {
          if (v < BN * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, n = bn0 + row;
            if (n < p.N && k0 < p.K) val = *reinterpret_cast<const uint4*>(Bp + (size_t)n * p.ldb + k0);
          }
        } else {
          if (v < BK * B_RV) {
            const int kk = v / B_RV, rc = v - kk * B_RV, k = kb + kk, n0 = bn0 + rc * VEC;
            if (k < p.K && n0 < p.N) val = *reinterpret_cast<const uint4*>(Bp + (size_t)k * p.ldb + n0);
          }
        }
        bv[i] = val;
      }
    };
    auto store_regs = [&]() {
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        const int v = tid + i * NT;
        if constexpr (!A_KM) {
          if (v < BM * KV) { const int row = v / KV, kc = v - row * KV; *reinterpret_cast<uint4*>(&As[row * C::LDK + kc * VEC]) = av[i]; }
        } else {
          if (v < BK * A_RV) { const int kk = v / A_RV, rc = v - kk * A_RV; lds_store_km<T>(&As[kk * C::LDRA + rc * VEC], av[i]); }
        }
      }
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int v = tid + i * NT;
        if constexpr (!B_KM) {
          if (v < BN * KV) { const int row = v / KV, kc = v - row * KV; *reinterpret_cast<uint4*>(&Bs[row * C::LDK + kc * VEC]) = bv[i]; }
        } else {
          if (v < BK * B_RV) { const int kk = v / B_RV, rc = v - kk * B_RV; lds_store_km<T>(&Bs[kk * C::LDRB + rc * VEC], bv[i]); }
        }
      }
    };
    if (nkt > 0) load_regs(0);
    for (int kt = 0; kt < nkt; ++kt) {
      store_regs();
      __syncthreads();
      if (kt + 1 < nkt) load_regs(kt + 1);
      tile_mma_step<C, T, A_KM, B_KM>(As, Bs, wm, wn, lane, acc);
      __syncthreads();
    }
  } else
  for (int kt = 0; kt < nkt; ++kt) {
    const int kb = kt * BK;
    // ---- A tile
    if constexpr (!A_KM) {
      constexpr int IT = (BM * KV + NT - 1) / NT;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int v = tid + i * NT;
        if (v < BM * KV) {
          const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, m = bm0 + row;
          T* d = &As[row * C::LDK + kc * VEC];
          if (vec_ok) {
            uint4 val = make_uint4(0, 0, 0, 0);
            if (m < p.M && k0 < p.K) val = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + k0);
            *reinterpret_cast<uint4*>(d) = val;
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] = (m < p.M && k0 + e < p.K) ? A[(size_t)m * p.lda + k0 + e] : (T)0;
          }
        }
      }
    } else {
      constexpr int RV = BM / VEC, IT = (BK * RV + NT - 1) / NT;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int v = tid + i * NT;
        if (v < BK * RV) {
          const int kk = v / RV, rc = v - kk * RV, k = kb + kk, m0 = bm0 + rc * VEC;
          T* d = &As[kk * C::LDRA + rc * VEC];
          if (vec_ok) {
            uint4 val = make_uint4(0, 0, 0, 0);
            if (k < p.K && m0 < p.M) val = *reinterpret_cast<const uint4*>(A + (size_t)k * p.lda + m0);
            lds_store_km<T>(d, val);
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] = (k < p.K && m0 + e < p.M) ? A[(size_t)k * p.lda + m0 + e] : (T)0;
          }
        }
      }
    }
    // ---- B tile
    if constexpr (!B_KM) {
      constexpr int IT = (BN * KV + NT - 1) / NT;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int v = tid + i * NT;
        if (v < BN * KV) {
          const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, n = bn0 + row;
          T* d = &Bs[row * C::LDK + kc * VEC];
          if (vec_ok) {
            uint4 val = make_uint4(0, 0, 0, 0);
            if (n < p.N && k0 < p.K) val = *reinterpret_cast<const uint4*>(Bp + (size_t)n * p.ldb + k0);
            *reinterpret_cast<uint4*>(d) = val;
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] = (n < p.N && k0 + e < p.K) ? Bp[(size_t)n * p.ldb + k0 + e] : (T)0;
          }
        }
      }
    } else {
      constexpr int RV = BN / VEC, IT = (BK * RV + NT - 1) / NT;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int v = tid + i * NT;
        if (v < BK * RV) {
          const int kk = v / RV, rc = v - kk * RV, k = kb + kk, n0 = bn0 + rc * VEC;
          T* d = &Bs[kk * C::LDRB + rc * VEC];
          if (vec_ok) {
            uint4 val = make_uint4(0, 0, 0, 0);
            if (k < p.K && n0 < p.N) val = *reinterpret_cast<const uint4*>(Bp + (size_t)k * p.ldb + n0);
            lds_store_km<T>(d, val);
          } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[e] = (k < p.K && n0 + e < p.N) ? Bp[(size_t)k * p.ldb + n0 + e] : (T)0;
          }
        }
      }
    }
    __syncthreads();
    tile_mma_step<C, T, A_KM, B_KM>(As, Bs, wm, wn, lane, acc);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < C::FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < C::FN; ++j) {
        const int n = bn0 + wn * WN + j * 16 + (lane & 15);
        if (n >= p.N) continue;
        float v = acc[i][j][r] * p.alpha;
        if (p.bias) v += p.bias[n];
        v = apply_act(v, p.act);
        const size_t o = coff + (size_t)m * p.ldc + n;
        if (p.c_f32) {
          float* c = reinterpret_cast<float*>(p.C);
          c[o] = (p.beta != 0.f) ? v + p.beta * c[o] : v;
        } else {
          T* c = reinterpret_cast<T*>(p.C);
          c[o] = ElemTraits<T>::from_f((p.beta != 0.f) ? v + p.beta * ElemTraits<T>::to_f(c[o]) : v);
        }
      }
    }
  }
}

// Small-problem variant (the fp32 planning head: M <= 780 rows, N, K <= 2048 -- forward, data-gradient and weight-gradient products of
// every Linear of the 6-layer decoder): 32 x 32 output tiles, and the FOUR waves of a workgroup split the K range of one tile between them
// (each wave has its own LDS stage and accumulators; the partial tiles are added in a fixed order through LDS at the end).  One launch,
// deterministic, >= 4x the workgroups of the 64 x 64 kernel and a quarter of its K-loop latency: the 64 x 64 kernel needed 20 us for
// 132 x 256 x 768 (12 workgroups x 24 K-steps), conv_gemm's split-K needed a second launch for the slice sum.
template <typename T, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void bgemm_ks_kernel(tfpp_bgemm_params p, int vec_ok) {
  using C = TileCfg<T, 32, 32, 32, 32>;
  constexpr int VEC = C::VEC, KV = C::KV, BK = C::BK, KW = 4, NL = 64;
  constexpr int A_ELEMS = A_KM ? C::A_ELEMS_KM : C::A_ELEMS_RM, B_ELEMS = B_KM ? C::B_ELEMS_KM : C::B_ELEMS_RM;
  constexpr int STAGE = A_ELEMS + B_ELEMS;
  static_assert(KW * STAGE * sizeof(T) >= KW * NL * 16 * sizeof(float), "the stages are reused for the cross-wave sum");
  __shared__ __attribute__((aligned(16))) T smem[KW * STAGE];
  const int lane = threadIdx.x & 63, kw = threadIdx.x >> 6;
  T* As = smem + kw * STAGE;
  T* Bs = As + A_ELEMS;
  const int z = blockIdx.z, z0 = z / p.batch1, z1 = z - z0 * p.batch1;
  const int bm0 = blockIdx.x * 32, bn0 = blockIdx.y * 32;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A) + z0 * p.a_bs0 + z1 * p.a_bs1;
  const T* __restrict__ Bp = reinterpret_cast<const T*>(p.B) + z0 * p.b_bs0 + z1 * p.b_bs1;
  const size_t coff = (size_t)z0 * p.c_bs0 + (size_t)z1 * p.c_bs1;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int nkt = (p.K + BK - 1) / BK, per = (nkt + KW - 1) / KW;
  if (vec_ok) {
    // aligned operands (round 6): register-prefetched loop, every wave on its own K range and its own LDS stage
    constexpr int RVA = 32 / VEC;
    constexpr int AIT = A_KM ? (BK * RVA + NL - 1) / NL : (32 * KV + NL - 1) / NL;
    constexpr int BIT = B_KM ? (BK * RVA + NL - 1) / NL : (32 * KV + NL - 1) / NL;
    uint4 av[AIT], bv[BIT];
    auto load_regs = [&](int kt) {
      const int kb = kt * BK;
      const bool live = kt < nkt;
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        const int v = lane + i * NL;
        uint4 val = make_uint4(0, 0, 0, 0);
        if constexpr (!A_KM) {
          if (live && v < 32 * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, m = bm0 + row;
            if (m < p.M && k0 < p.K) val = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + k0);
          }
        } else {
          if (live && v < BK * RVA) {
            const int kk = v / RVA, rc = v - kk * RVA, k = kb + kk, m0 = bm0 + rc * VEC;
            if (k < p.K && m0 < p.M) val = *reinterpret_cast<const uint4*>(A + (size_t)k * p.lda + m0);
          }
        }
        av[i] = val;
      }
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int v = lane + i * NL;
        uint4 val = make_uint4(0, 0, 0, 0);
        if constexpr (!B_KM) {
          if (live && v < 32 * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, n = bn0 + row;
            if (n < p.N && k0 < p.K) val = *reinterpret_cast<const uint4*>(Bp + (size_t)n * p.ldb + k0);
          }
        } else {
          if (live && v < BK * RVA) {
            const int kk = v / RVA, rc = v - kk * RVA, k = kb + kk, n0 = bn0 + rc * VEC;
            if (k < p.K && n0 < p.N) val = *reinterpret_cast<const uint4*>(Bp + (size_t)k * p.ldb + n0);
          }
        }
        bv[i] = val;
      }
    };
    auto store_regs = [&]() {
#pragma unroll
      for (int i = 0; i < AIT; ++i) {
        const int v = lane + i * NL;
        if constexpr (!A_KM) {
          if (v < 32 * KV) { const int row = v / KV, kc = v - row * KV; *reinterpret_cast<uint4*>(&As[row * C::LDK + kc * VEC]) = av[i]; }
        } else {
          if (v < BK * RVA) { const int kk = v / RVA, rc = v - kk * RVA; lds_store_km<T>(&As[kk * C::LDRA + rc * VEC], av[i]); }
        }
      }
#pragma unroll
      for (int i = 0; i < BIT; ++i) {
        const int v = lane + i * NL;
        if constexpr (!B_KM) {
          if (v < 32 * KV) { const int row = v / KV, kc = v - row * KV; *reinterpret_cast<uint4*>(&Bs[row * C::LDK + kc * VEC]) = bv[i]; }
        } else {
          if (v < BK * RVA) { const int kk = v / RVA, rc = v - kk * RVA; lds_store_km<T>(&Bs[kk * C::LDRB + rc * VEC], bv[i]); }
        }
      }
    };
    if (per > 0) load_regs(kw * per);
    for (int it = 0; it < per; ++it) {
      const int kt = kw * per + it;
      store_regs();  // (a tile past the wave's range is stored as zeros and multiplied: exact)
      __syncthreads();
      if (it + 1 < per) load_regs(kt + 1);
      if (kt < nkt) tile_mma_step<C, T, A_KM, B_KM>(As, Bs, 0, 0, lane, acc);
      __syncthreads();
    }
  } else
  for (int it = 0; it < per; ++it) {  // (the same trip count in every wave: the barriers below are workgroup barriers)
    const int kt = kw * per + it;
    const bool live = kt < nkt;
    const int kb = kt * BK;
    if (live) {
      if constexpr (!A_KM) {
        constexpr int IT = (32 * KV + NL - 1) / NL;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const int v = lane + i * NL;
          if (v < 32 * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, m = bm0 + row;
            T* d = &As[row * C::LDK + kc * VEC];
            if (vec_ok) {
              uint4 val = make_uint4(0, 0, 0, 0);
              if (m < p.M && k0 < p.K) val = *reinterpret_cast<const uint4*>(A + (size_t)m * p.lda + k0);
              *reinterpret_cast<uint4*>(d) = val;
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) d[e] = (m < p.M && k0 + e < p.K) ? A[(size_t)m * p.lda + k0 + e] : (T)0;
            }
          }
        }
      } else {
        constexpr int RV = 32 / VEC, IT = (BK * RV + NL - 1) / NL;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const int v = lane + i * NL;
          if (v < BK * RV) {
            const int kk = v / RV, rc = v - kk * RV, k = kb + kk, m0 = bm0 + rc * VEC;
            T* d = &As[kk * C::LDRA + rc * VEC];
            if (vec_ok) {
              uint4 val = make_uint4(0, 0, 0, 0);
              if (k < p.K && m0 < p.M) val = *reinterpret_cast<const uint4*>(A + (size_t)k * p.lda + m0);
              lds_store_km<T>(d, val);
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) d[e] = (k < p.K && m0 + e < p.M) ? A[(size_t)k * p.lda + m0 + e] : (T)0;
            }
          }
        }
      }
      if constexpr (!B_KM) {
        constexpr int IT = (32 * KV + NL - 1) / NL;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const int v = lane + i * NL;
          if (v < 32 * KV) {
            const int row = v / KV, kc = v - row * KV, k0 = kb + kc * VEC, n = bn0 + row;
            T* d = &Bs[row * C::LDK + kc * VEC];
            if (vec_ok) {
              uint4 val = make_uint4(0, 0, 0, 0);
              if (n < p.N && k0 < p.K) val = *reinterpret_cast<const uint4*>(Bp + (size_t)n * p.ldb + k0);
              *reinterpret_cast<uint4*>(d) = val;
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) d[e] = (n < p.N && k0 + e < p.K) ? Bp[(size_t)n * p.ldb + k0 + e] : (T)0;
            }
          }
        }
      } else {
        constexpr int RV = 32 / VEC, IT = (BK * RV + NL - 1) / NL;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          const int v = lane + i * NL;
          if (v < BK * RV) {
            const int kk = v / RV, rc = v - kk * RV, k = kb + kk, n0 = bn0 + rc * VEC;
            T* d = &Bs[kk * C::LDRB + rc * VEC];
            if (vec_ok) {
              uint4 val = make_uint4(0, 0, 0, 0);
              if (k < p.K && n0 < p.N) val = *reinterpret_cast<const uint4*>(Bp + (size_t)k * p.ldb + n0);
              lds_store_km<T>(d, val);
            } else {
#pragma unroll
              for (int e = 0; e < VEC; ++e) d[e] = (k < p.K && n0 + e < p.N) ? Bp[(size_t)k * p.ldb + n0 + e] : (T)0;
            }
          }
        }
      }
    }
    __syncthreads();
    if (live) tile_mma_step<C, T, A_KM, B_KM>(As, Bs, 0, 0, lane, acc);
    __syncthreads();
  }
  // cross-wave sum in a fixed order (wave 0 + 1 + 2 + 3), then the epilogue of bgemm_kernel by wave 0
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((kw * 16) + (i * 2 + j) * 4 + r) * NL + lane] = acc[i][j][r];
  __syncthreads();
  if (kw != 0) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm0 + i * 16 + (lane >> 4) * 4 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = bn0 + j * 16 + (lane & 15);
        if (n >= p.N) continue;
        const int e = (i * 2 + j) * 4 + r;
        float v = ((red[e * NL + lane] + red[(16 + e) * NL + lane]) + red[(32 + e) * NL + lane]) + red[(48 + e) * NL + lane];
        v *= p.alpha;
        if (p.bias) v += p.bias[n];
        v = apply_act(v, p.act);
        const size_t o = coff + (size_t)m * p.ldc + n;
        if (p.c_f32) {
          float* c = reinterpret_cast<float*>(p.C);
          c[o] = (p.beta != 0.f) ? v + p.beta * c[o] : v;
        } else {
          T* c = reinterpret_cast<T*>(p.C);
          c[o] = ElemTraits<T>::from_f((p.beta != 0.f) ? v + p.beta * ElemTraits<T>::to_f(c[o]) : v);
        }
      }
    }
  }
}

// which kernel tfpp_bgemm runs: 1 = the small-problem kernel above (fp32, few 64 x 64 tiles, a K loop worth splitting), 0 = 64 x 64 tiles
static int bgemm_variant(const tfpp_bgemm_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_BGEMM_KS"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || dtype != TFPP_F32) return 0;
  const long tiles64 = (long)cdiv(p.M, 64) * cdiv(p.N, 64) * p.batch0 * p.batch1;
  return (tiles64 <= 256 && p.K >= 96) ? 1 : 0;
}
extern "C" int tfpp_bgemm_variant(const tfpp_bgemm_params* p, int dtype) { return p ? bgemm_variant(*p, dtype) : TFPP_EINVAL; }

template <typename T, bool A_KM, bool B_KM> static int launch_bgemm(const tfpp_bgemm_params& p, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  auto al = [&](long v) { return (v % VEC) == 0; };
  int vec_ok = al(p.lda) && al(p.ldb) && al(p.a_bs0) && al(p.a_bs1) && al(p.b_bs0) && al(p.b_bs1) &&
               (((uintptr_t)p.A & 15) == 0) && (((uintptr_t)p.B & 15) == 0);
  vec_ok = vec_ok && (A_KM ? al(p.M) : al(p.K)) && (B_KM ? al(p.N) : al(p.K));
  using C = TileCfg<T, 64, 64, 32, 32>;
  if constexpr (sizeof(T) == 4) {
    if (bgemm_variant(p, TFPP_F32) == 1) {
      hipLaunchKernelGGL((bgemm_ks_kernel<T, A_KM, B_KM>), dim3(cdiv(p.M, 32), cdiv(p.N, 32), p.batch0 * p.batch1), dim3(256), 0, st, p, vec_ok);
      TFPP_CHECK_LAUNCH();
      return 0;
    }
  }
  dim3 grid(cdiv(p.M, 64), cdiv(p.N, 64), p.batch0 * p.batch1);
  hipLaunchKernelGGL((bgemm_kernel<T, 64, 64, 32, 32, A_KM, B_KM>), grid, dim3(C::NT), 0, st, p, vec_ok);
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <typename T> static int dispatch_bgemm(const tfpp_bgemm_params& p, hipStream_t st) {
  if (p.a_km) return p.b_km ? launch_bgemm<T, true, true>(p, st) : launch_bgemm<T, true, false>(p, st);
  return p.b_km ? launch_bgemm<T, false, true>(p, st) : launch_bgemm<T, false, false>(p, st);
}

extern "C" int tfpp_bgemm(const tfpp_bgemm_params* p, int dtype, void* stream) {
  if (!p || !p->A || !p->B || !p->C || p->batch0 < 1 || p->batch1 < 1) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) return dispatch_bgemm<float>(*p, st);
  if (dtype == TFPP_BF16) return dispatch_bgemm<bf16_t>(*p, st);
  return TFPP_EINVAL;
}
