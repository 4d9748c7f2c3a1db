// MFMA tile core shared by the implicit-GEMM convolution, weight-gradient and batched GEMM kernels.
//
// gfx950 MFMA fragments used (wave64):
//   bf16: v_mfma_f32_16x16x32_bf16   A: lane l holds A[i=l&15][k=(l>>4)*8 .. +7]   B: B[k=(l>>4)*8..+7][j=l&15]
//   f32 : v_mfma_f32_16x16x4_f32     A: lane l holds A[i=l&15][k=l>>4]             B: B[k=l>>4][j=l&15]
//   C/D (both): acc[r] = D[i=(l>>4)*4+r][j=l&15]
// A K-step is 32 for both types.  For f32 a lane still reads 8 consecutive k (two 16-byte LDS reads) and issues
// 8 MFMAs, the j-th using element j of both operands: MFMA j then reduces k = {8g+j : g=0..3}, the same set for A
// and B, so the sum over all 8 covers k=0..31 exactly once (order of fp32 accumulation differs from a CPU loop
// only by this permutation).
//
// LDS images: "row-major" [row][k] (pitch LDK, 16-byte aligned rows, fragment = 16-byte reads) for operands whose
// global layout is k-contiguous, and "k-major" [k][row] (pitch LDR) for operands that are row-contiguous in HBM
// (weight-gradient GEMMs, P@V): those are stored as loaded (8-byte LDS writes) and gathered element-wise into
// fragments, which transposes them without a global-memory pass.
#pragma once
#include "common.h"
#include "../../include/tfpp.h"

template <typename T, int BM_, int BN_, int WM_, int WN_, int BK_ = 32>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_;  // BK: K elements per LDS stage (multiple of 32)
  static constexpr int VEC = ElemTraits<T>::VEC;
  static constexpr int KV = BK / VEC;                       // 16-byte vectors per tile row
  static constexpr int LDK = BK + (sizeof(T) == 2 ? 8 : 4);  // row-major pitch (elements): 80 B / 144 B
  static constexpr int PADR = (sizeof(T) == 2 ? 16 : 2);     // k-major pitch pad (bf16: 32-byte row skew for ds_read_tr16_b64)
  static constexpr int WAVES_M = BM / WM, WAVES_N = BN / WN;
  static constexpr int NT = WAVES_M * WAVES_N * TFPP_WAVE;
  static constexpr int FM = WM / 16, FN = WN / 16;
  static constexpr int LDRA = BM + PADR, LDRB = BN + PADR;
  static constexpr int A_ELEMS_RM = BM * LDK, B_ELEMS_RM = BN * LDK;
  static constexpr int A_ELEMS_KM = BK * LDRA, B_ELEMS_KM = BK * LDRB;
};

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { uint4 v; };
template <> struct Frag<float> { float v[8]; };

// fragment from a row-major LDS image: p points at element [row = frag_row0 + (l&15)][k = (l>>4)*8]
__device__ __forceinline__ void frag_load_rm(Frag<bf16_t>& f, const bf16_t* p) { f.v = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void frag_load_rm(Frag<float>& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// fragment from a k-major LDS image [k][row] (pitch ldr): p points at element [k = (l>>4)*8][row = frag_row0], r16 = l&15.
// bf16: two hardware transpose reads (ds_read_b64_tr_b16).  Within each 16-lane group, lane m supplies the address of 4
// consecutive rows at k-row (m>>2), row chunk (m&3)*4; the instruction returns to lane i the 4 k-values of row i
// (out[i][j] = in[4j + (i>>2)][i&3] -- measured on gfx950).  Two reads give k = 0..3 and 4..7.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ void frag_load_km(Frag<bf16_t>& f, const bf16_t* p, int ldr, int r16) {
  const bf16_t* a0 = p + (r16 >> 2) * ldr + (r16 & 3) * 4;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a0 + 4 * ldr));
  f.v = make_uint4((unsigned)(unsigned short)v0[0] | ((unsigned)(unsigned short)v0[1] << 16),
                   (unsigned)(unsigned short)v0[2] | ((unsigned)(unsigned short)v0[3] << 16),
                   (unsigned)(unsigned short)v1[0] | ((unsigned)(unsigned short)v1[1] << 16),
                   (unsigned)(unsigned short)v1[2] | ((unsigned)(unsigned short)v1[3] << 16));
}
__device__ __forceinline__ void frag_load_km(Frag<float>& f, const float* p, int ldr, int r16) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = p[j * ldr + r16];
}

__device__ __forceinline__ void frag_mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x4_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a.v), __builtin_bit_cast(bf16x8_t, b.v), acc, 0, 0,
                                                0);
}
__device__ __forceinline__ void frag_mma(const Frag<float>& a, const Frag<float>& b, f32x4_t& acc) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// All K-steps (BK/32 of them) of the block tile held in LDS.
template <typename C, typename T, bool A_KM, bool B_KM>
__device__ __forceinline__ void tile_mma_step(const T* As, const T* Bs, int wm, int wn, int lane, f32x4_t (&acc)[C::FM][C::FN]) {
  const int r16 = lane & 15;
#pragma unroll
 for (int ks = 0; ks < C::BK / 32; ++ks) {
  Frag<T> fa[C::FM], fb[C::FN];
  const int kg = ks * 32 + (lane >> 4) * 8;
#pragma unroll
  for (int i = 0; i < C::FM; ++i) {
    const int row0 = wm * C::WM + i * 16;
    if constexpr (A_KM) frag_load_km(fa[i], As + kg * C::LDRA + row0, C::LDRA, r16);
    else frag_load_rm(fa[i], As + (row0 + r16) * C::LDK + kg);
  }
#pragma unroll
  for (int j = 0; j < C::FN; ++j) {
    const int row0 = wn * C::WN + j * 16;
    if constexpr (B_KM) frag_load_km(fb[j], Bs + kg * C::LDRB + row0, C::LDRB, r16);
    else frag_load_rm(fb[j], Bs + (row0 + r16) * C::LDK + kg);
  }
#pragma unroll
  for (int i = 0; i < C::FM; ++i)
#pragma unroll
    for (int j = 0; j < C::FN; ++j) frag_mma(fa[i], fb[j], acc[i][j]);
 }
}

// store a 16-byte vector (VEC elements along rows) into a k-major image at [k][row0..row0+VEC) -- 8-byte aligned
template <typename T> __device__ __forceinline__ void lds_store_km(T* dst, const uint4& v) {
  uint2* d = reinterpret_cast<uint2*>(dst);
  d[0] = make_uint2(v.x, v.y);
  d[1] = make_uint2(v.z, v.w);
}

// ---------------------------------------------------------------------------------------------------------------
// Coalesced bf16 epilogue.  The MFMA C layout gives a lane 4 rows x 1 column per fragment, so storing straight from the
// accumulators issues 2-byte stores that touch 4 cache lines per instruction with 32 valid bytes each; the phase trace
// of the LDS-DMA kernel (tools/glds_trace.py) showed that costing 6-7 us per workgroup -- as much as an 18-tile K loop.
// Instead each wave passes 16 rows x (16 FN) columns of fp32 results through a private LDS strip and every lane then
// finishes 8 consecutive channels of one row: scale / shift / residual are 16-byte loads, the store is 16 bytes, and a
// wave writes 64..128-byte row segments.  Preconditions (epi_vec_ok): NHWC bf16 destination, n_g, dst_ld (res_ld)
// multiples of 8, 16-byte aligned bases.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool epi_vec_ok(const tfpp_conv_params& p) {
  return !p.dst_nchw && !p.dst_f32 && ((p.n_g | (int)p.dst_ld) & 7) == 0 && ((uintptr_t)p.dst & 15) == 0 &&
         (!p.res || (((int)p.res_ld & 7) == 0 && ((uintptr_t)p.res & 15) == 0));
}
template <int FN> struct EpiStrip { static constexpr int PITCH = FN * 16 + 4, FLOATS = 16 * PITCH; };

// Fused BatchNorm-backward statistics (tfpp_conv_params.bns_*): in the vector epilogue a lane always finishes the SAME 8 channels
// (chunk lane % (2 FN) of its wave's column strip), so it keeps the two sums of those channels in registers over all passes.
// The y / x operands of every pass are fetched up front (prefetch()): issued back to back they cost one memory latency per
// workgroup; loaded inside the passes they cost one per pass (the first version made the data-gradient GEMMs ~10 % slower).
template <int FN, int FM> struct BnsAcc {
  static constexpr int CH = FN * 2, NCHUNK = 16 * CH, CI = (NCHUNK + 63) / 64;
  float s0[8], s1[8], mu[8], is[8];
  uint4 yq[FM][CI], xq[FM][CI];
  __device__ __forceinline__ void init(const tfpp_conv_params& p, int lane, int n_base, int g) {
    const int n = n_base + (lane % CH) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; }
    if (p.bns_partial && n < p.n_g) {
      const int ch = g * p.n_g + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) { mu[e] = p.bns_mean[ch + e]; is[e] = p.bns_invstd[ch + e]; }
    }
  }
  // operands of pass i (rows m_pass .. m_pass + rows_valid - 1): same lane -> (row, chunk) map as epi_pass_bf16
  __device__ __forceinline__ void prefetch(const tfpp_conv_params& p, int lane, int i, long m_pass, int rows_valid, int n_base, int g) {
#pragma unroll
    for (int c = 0; c < CI; ++c) {
      const int q = lane + c * 64, row = q / CH, n = n_base + (q - row * CH) * 8;
      yq[i][c] = make_uint4(0, 0, 0, 0);
      xq[i][c] = make_uint4(0, 0, 0, 0);
      if (q < NCHUNK && row < rows_valid && n < p.n_g) {
        const size_t off = (size_t)(m_pass + row) * p.bns_ld + g * p.n_g + n;
        xq[i][c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bns_x) + off);
        if (p.bns_relu) yq[i][c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.bns_y) + off);
      }
    }
  }
  // gv: the 8 finished results (already rounded to bf16 and unpacked) of chunk c of pass i
  __device__ __forceinline__ void add(const tfpp_conv_params& p, const float* gv, int i, int c) {
    float xv[8], g[8];
    unpack16<bf16_t>(xq[i][c], xv);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gv[e];
    if (p.bns_relu) {
      float yv[8];
      unpack16<bf16_t>(yq[i][c], yv);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * (xv[e] - mu[e]) * is[e]; }
  }
  // Combine the lanes of a wave, then the WGM waves that share a column strip, and store row `mtile` of bns_partial.
  // sm: WGM * WGN * 2 FN * 16 floats of LDS nobody else is using; contains a __syncthreads (call from uniform control flow).
  template <int WGM, int WGN>
  __device__ __forceinline__ void finish(const tfpp_conv_params& p, float* sm, int wm, int wn, int lane, int mtile, int n_base, int g) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = CH; o < 64; o <<= 1) { s0[e] += __shfl_xor(s0[e], o, 64); s1[e] += __shfl_xor(s1[e], o, 64); }
    }
    if (lane < CH) {
      float* d = sm + ((wm * WGN + wn) * CH + lane) * 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) { d[e] = s0[e]; d[8 + e] = s1[e]; }
    }
    __syncthreads();
    if (wm == 0 && lane < CH) {
      const int n = n_base + lane * 8, ctot = p.G * p.n_g;
      if (n < p.n_g) {
        float a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
        for (int w2 = 0; w2 < WGM; ++w2) {
          const float* d = sm + ((w2 * WGN + wn) * CH + lane) * 16;
#pragma unroll
          for (int e = 0; e < 16; ++e) a[e] += d[e];
        }
        float* row = p.bns_partial + (size_t)mtile * 2 * ctot + g * p.n_g + n;
#pragma unroll
        for (int e = 0; e < 8; ++e) { row[e] = a[e]; row[ctot + e] = a[8 + e]; }
      }
    }
  }
};

// one pass: rows m_pass .. m_pass + rows_valid - 1 (<= 16), columns n_base .. n_base + 16 FN - 1 of group g
// BNS is a template parameter and the callers branch on p.bns_partial around two instantiations: with a run-time `bns` pointer (null or
// not) the BnsAcc object could not be kept in registers and EVERY bf16 conv launch paid 128 B/thread of scratch initialisation (2-3x the
// bytes of C in WRITE_SIZE, profiles/r02_pmc_calibration.txt).
template <int FN, int FM = 1, bool BNS = false>
__device__ __forceinline__ void epi_pass_bf16(const tfpp_conv_params& p, const f32x4_t (&acc)[FN], float* strip, int lane, long m_pass,
                                              int rows_valid, int n_base, int g, BnsAcc<FN, FM>* bns = nullptr, int ipass = 0) {
  constexpr int PITCH = EpiStrip<FN>::PITCH, CH = FN * 2, NCHUNK = 16 * CH;
  const int p16 = lane & 15, kg = lane >> 4;
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) strip[(kg * 4 + r) * PITCH + j * 16 + p16] = acc[j][r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private strip: DS ops retire in order, this also pins the compiler
#pragma unroll
  for (int c = 0; c < (NCHUNK + 63) / 64; ++c) {
    const int q = lane + c * 64;
    const int row = q / CH, c8 = q - row * CH, n = n_base + c8 * 8;
    if (q < NCHUNK && row < rows_valid && n < p.n_g) {
      const float4 lo = *reinterpret_cast<const float4*>(strip + row * PITCH + c8 * 8);
      const float4 hi = *reinterpret_cast<const float4*>(strip + row * PITCH + c8 * 8 + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const int ch = g * p.n_g + n;
      const long m = m_pass + row;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      if (p.scale) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.scale + ch), s1 = *reinterpret_cast<const float4*>(p.scale + ch + 4);
        v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w; v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      if (p.shift) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.shift + ch), s1 = *reinterpret_cast<const float4*>(p.shift + ch + 4);
        v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
      }
      if (p.res) {
        float rv[8];
        load_vec<bf16_t>(reinterpret_cast<const bf16_t*>(p.res) + (size_t)m * p.res_ld + ch, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
      }
      if (p.relu_mask) {  // ReLU backward of the tensor this gradient belongs to (tfpp.h)
        float fv[8];
        load_vec<bf16_t>(reinterpret_cast<const bf16_t*>(p.relu_mask) + (size_t)m * p.relu_mask_ld + ch, fv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fv[e] > 0.f ? v[e] : 0.f;
      }
      const uint4 packed = pack16<bf16_t>(v);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.dst) + (size_t)m * p.dst_ld + ch) = packed;
      if constexpr (BNS) {
        float gv[8];
        unpack16<bf16_t>(packed, gv);  // the rounded values: exactly what the BatchNorm backward reads back
        bns->add(p, gv, ipass, c);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip is rewritten by the next pass
}

// Second half of epi_pass_bf16 on its own (the strip already holds 16 rows x 16 FN columns of fp32 results): used by kernels whose
// accumulator layout is not the 16x16 one (32x32x16 MFMA tiles).  Same lane -> (row, 8-channel chunk) map, same arithmetic.
template <int FN>
__device__ __forceinline__ void epi_finish_bf16(const tfpp_conv_params& p, const float* strip, int lane, long m_pass, int rows_valid,
                                                int n_base, int g) {
  constexpr int PITCH = EpiStrip<FN>::PITCH, CH = FN * 2, NCHUNK = 16 * CH;
#pragma unroll
  for (int c = 0; c < (NCHUNK + 63) / 64; ++c) {
    const int q = lane + c * 64;
    const int row = q / CH, c8 = q - row * CH, n = n_base + c8 * 8;
    if (q < NCHUNK && row < rows_valid && n < p.n_g) {
      const float4 lo = *reinterpret_cast<const float4*>(strip + row * PITCH + c8 * 8);
      const float4 hi = *reinterpret_cast<const float4*>(strip + row * PITCH + c8 * 8 + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      const int ch = g * p.n_g + n;
      const long m = m_pass + row;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      if (p.scale) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.scale + ch), s1 = *reinterpret_cast<const float4*>(p.scale + ch + 4);
        v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w; v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      if (p.shift) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.shift + ch), s1 = *reinterpret_cast<const float4*>(p.shift + ch + 4);
        v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
      }
      if (p.res) {
        float rv[8];
        load_vec<bf16_t>(reinterpret_cast<const bf16_t*>(p.res) + (size_t)m * p.res_ld + ch, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (p.act != ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act);
      }
      if (p.relu_mask) {
        float fv[8];
        load_vec<bf16_t>(reinterpret_cast<const bf16_t*>(p.relu_mask) + (size_t)m * p.relu_mask_ld + ch, fv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fv[e] > 0.f ? v[e] : 0.f;
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.dst) + (size_t)m * p.dst_ld + ch) = pack16<bf16_t>(v);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip is rewritten by the next pass
}
