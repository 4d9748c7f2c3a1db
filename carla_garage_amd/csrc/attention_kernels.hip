// Fused multi-head self-attention of the fusion transformers (team_code/transfuser.py:362-380), bf16 storage, fp32 accumulation:
//     O = dropout(softmax(scale * Q K^T)) V        S = 320 tokens (256 image + 64 LiDAR anchors), 4 heads, d = 24/56/144/384 (padded)
// The reference runs matmul -> softmax -> dropout -> matmul and materialises the 320x320 scores per head; the unfused HIP path did
// the same in three launches (bgemm, softmax, bgemm) plus five in backward.  Here the scores never leave the registers:
//
// forward, one workgroup = (batch, head, 64 queries), 4 waves x 16 queries:
//   1. S^T = K Q^T ("swapped" product): MFMA A operand = K rows (keys), B operand = Q rows.  In the 16x16 C layout a lane then
//      owns ONE query (column l & 15) and, over the 20 key fragments, 80 of its 320 keys -- the other 240 sit in the three lanes
//      l ^ 16, l ^ 32, l ^ 48.  K is staged per 32-wide d slice through a double-buffered LDS image shared by the 4 waves.
//   2. softmax over the keys = 80 register values + two wave shuffles (xor 16, xor 32); log-sum-exp saved for backward; attention
//      dropout from the same counter-based hash and element index (row * T + key) as the unfused tfpp_softmax_fwd.
//   3. O = P V: the C fragments of two neighbouring key blocks ARE an MFMA A fragment (16 queries x 32 keys) once the contraction
//      slots are labelled  slot e of lane group kg -> key 4 kg + e (e < 4) / 16 + 4 kg + e - 4  inside a 32-key block: no data
//      movement, only a bf16 pack.  The B operand (V^T) comes from an LDS image of V [key][128-wide d chunk] through
//      ds_read_b64_tr_b16 addressed with the same labelling (16-byte chunks XOR-swizzled by (key & 7) << 1: conflict-free).
//      O leaves through a per-wave LDS strip as 16-byte row segments.
// backward: see attn_bwd_* below (recomputes P from Q, K and the saved log-sum-exp; dQ in one launch, dK / dV in another).
#include "gemm_core.h"
#include <cstdlib>

namespace {
constexpr int ATT_TQ = 64;       // queries per workgroup
constexpr int ATT_KF = 20;       // key fragments of 16 -> T <= 320
constexpr int ATT_DC = 128;      // d chunk of the V image
constexpr int K_SLICE_BYTES = ATT_KF * 16 * 64;           // K image: [320 keys][32 d] bf16 = 20 KB, double-buffered
constexpr int V_IMG_BYTES = ATT_KF * 16 * ATT_DC * 2;     // V image: [320 keys][128 d] bf16 = 80 KB
constexpr int O_STRIP_BYTES = 16 * ATT_DC * 2;            // per wave: 16 queries x 128 d bf16 = 4 KB (aliases the K image)

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) { return f2bf_pack2(a, b); }

// K image: 64-byte rows, chunk kc (16 B) of key row r at r * 64 + ((kc ^ (r & 3)) * 16): the 16 rows of a ds_read_b128 lane group
// (4 rows per 256-byte bank row) land on distinct 16-byte slots
__device__ __forceinline__ int kimg_off(int row, int kc) { return row * 64 + ((kc ^ (row & 3)) * 16); }
// V image: 256-byte rows, chunk c of key row r at r * 256 + ((c ^ ((r & 7) << 1)) * 16)
__device__ __forceinline__ int vimg_off(int row, int c) { return row * 256 + ((c ^ ((row & 7) << 1)) * 16); }

struct AttnGeo {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  long ld_q, ld_kv;
};

// S^T fragments of one wave: acc[f][r] = scale-free score of key f*16 + (lane>>4)*4 + r and query q_row0 + (lane & 15)
template <int NKS>
__device__ __forceinline__ void attn_scores(const bf16_t* __restrict__ qh, long ld_q, const bf16_t* __restrict__ kh, long ld_kv, int T, int d,
                                            int q_row, unsigned char* kimg, int tid, int lane, f32x4_t (&S)[ATT_KF]) {
  const int p16 = lane & 15, kg = lane >> 4;
  // this lane's Q operand (B fragment): 8 consecutive d of its query, one 16-byte load per k-step
  uint4 qf[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int d0 = ks * 32 + kg * 8;
    qf[ks] = (d0 < d) ? *reinterpret_cast<const uint4*>(qh + (size_t)q_row * ld_q + d0) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f) S[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // K slice staging: T * 4 chunks of 16 bytes per k-step, 256 threads -> 5 chunks per thread, prefetched PF k-steps ahead (a k-step
  // is only 20 MFMAs per wave: one L2 latency per k-step was the whole cost of this phase with a one-deep prefetch)
  constexpr int KIT = ATT_KF * 16 * 4 / 256;
  constexpr int PF = NKS < 3 ? NKS : 3;
  uint4 kreg[PF][KIT];
  auto kload = [&](int ks, uint4 (&dst)[KIT]) {
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int c = tid + it * 256, row = c >> 2, kc = c & 3, d0 = ks * 32 + kc * 8;
      dst[it] = (row < T && d0 < d) ? *reinterpret_cast<const uint4*>(kh + (size_t)row * ld_kv + d0) : make_uint4(0, 0, 0, 0);
    }
  };
#pragma unroll
  for (int ks = 0; ks < PF; ++ks) kload(ks, kreg[ks]);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    unsigned char* img = kimg + (ks & 1) * K_SLICE_BYTES;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int c = tid + it * 256, row = c >> 2, kc = c & 3;
      *reinterpret_cast<uint4*>(img + kimg_off(row, kc)) = kreg[ks % PF][it];
    }
    __syncthreads();  // slice ks visible; every wave is past its reads of slice ks - 1 (the buffer written next iteration)
    if (ks + PF < NKS) kload(ks + PF, kreg[ks % PF]);
    // rows >= T of the image are zero (kload), so the fragments past the sequence contribute exact zeros: no predicates in the
    // MFMA loop (a uniform branch per fragment kept the compiler from batching the 20 LDS reads in front of the MFMAs)
    Frag<bf16_t> a[ATT_KF];
#pragma unroll
    for (int f = 0; f < ATT_KF; ++f) a[f].v = *reinterpret_cast<const uint4*>(img + kimg_off(f * 16 + p16, kg));
    Frag<bf16_t> b;
    b.v = qf[ks];
#pragma unroll
    for (int f = 0; f < ATT_KF; ++f) frag_mma(a[f], b, S[f]);
  }
}

// scores -> probabilities in place (fp32), returns the log-sum-exp of the lane's query.  invalid key fragments hold 0.
__device__ __forceinline__ float attn_softmax(f32x4_t (&S)[ATT_KF], int nkf, float scale) {
  float mx = -3.0e38f;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[f][r] *= scale; mx = fmaxf(mx, S[f][r]); }
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[f][r] = __expf(S[f][r] - mx); sum += S[f][r]; }
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[f][r] *= inv;
    }
  return mx + __logf(sum);
}

// window attention: scores -> probabilities with the dense relative-position bias and shift mask added; keys >= T are excluded.
// bias / mask rows of this lane's query (pitch-ld_b rows, 16-byte aligned: the lane's 4 keys of a fragment are one float4)
__device__ __forceinline__ void attn_softmax_window(f32x4_t (&S)[ATT_KF], int nkf, int T, int kg, float scale, const float* __restrict__ brow,
                                                    const float* __restrict__ mrow) {
  float mx = -3.0e38f;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
      const int key0 = f * 16 + kg * 4;
      const float4 bv = *reinterpret_cast<const float4*>(brow + key0);
      float4 mv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mrow) mv = *reinterpret_cast<const float4*>(mrow + key0);
      const float add[4] = {bv.x + mv.x, bv.y + mv.y, bv.z + mv.z, bv.w + mv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        S[f][r] = (key0 + r < T) ? S[f][r] * scale + add[r] : -3.0e38f;
        mx = fmaxf(mx, S[f][r]);
      }
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { S[f][r] = __expf(S[f][r] - mx); sum += S[f][r]; }
    }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
    if (f < nkf) {
#pragma unroll
      for (int r = 0; r < 4; ++r) S[f][r] *= inv;
    }
}

// stage one 128-wide d chunk [dc, dc + 128) of a [T][*] token matrix (row stride ld) into the swizzled V image; chunks past d are zero.
// The loads go out in batches of 10 per thread before the first LDS store (a load -> store -> load chain costs one L2 latency per
// 16 bytes: the first version of this kernel spent most of its time here).
__device__ __forceinline__ void stage_vimg(const bf16_t* __restrict__ vh, long ld, int T, int d, int dc, int cpr, unsigned char* vimg, int tid,
                                           int t_pad = 0) {
  // rows T .. t_pad - 1 (window attention: T = 147 inside 32-key contraction blocks) are written as zeros: their probabilities are exact
  // zeros, but 0 x stale LDS contents could be NaN
  const int total = (t_pad > T ? t_pad : T) * cpr;  // cpr: 16-byte chunks per row actually used (2 per 16-wide output fragment; a power of two in this model)
  const int sh = __builtin_ctz(cpr);
  const bool pow2 = (cpr & (cpr - 1)) == 0;
  constexpr int U = 10;
  for (int c0 = tid; c0 < total; c0 += 256 * U) {
    uint4 r[U];
    int off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * 256;
      r[u] = make_uint4(0, 0, 0, 0);
      off[u] = -1;
      if (c < total) {
        const int row = pow2 ? (c >> sh) : (c / cpr), ch = c - row * cpr, d0 = dc + ch * 8;
        off[u] = vimg_off(row, ch);
        if (d0 < d && row < T) r[u] = *reinterpret_cast<const uint4*>(vh + (size_t)row * ld + d0);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (off[u] >= 0) *reinterpret_cast<uint4*>(vimg + off[u]) = r[u];
  }
}

// B fragment (16 columns n0 .. n0+15 of the image, 32 keys of block kk in the slot labelling above) by two transpose reads
__device__ __forceinline__ void vimg_frag(const unsigned char* vimg, int kk, int nfrag, int lane, Frag<bf16_t>& f) {
  const int m16 = lane & 15, kg = lane >> 4;
  const int row0 = kk * 32 + 4 * kg + (m16 >> 2);
  const int c = nfrag * 2 + ((m16 & 3) >> 1), half = (m16 & 1) * 8;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vimg + vimg_off(row0, c) + half));
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(vimg + vimg_off(row0 + 16, c) + half));
  f.v = make_uint4((unsigned)(unsigned short)v0[0] | ((unsigned)(unsigned short)v0[1] << 16),
                   (unsigned)(unsigned short)v0[2] | ((unsigned)(unsigned short)v0[3] << 16),
                   (unsigned)(unsigned short)v1[0] | ((unsigned)(unsigned short)v1[1] << 16),
                   (unsigned)(unsigned short)v1[2] | ((unsigned)(unsigned short)v1[3] << 16));
}

// o = A (fragments af: ATT_KF/2 contraction blocks of 32) x image chunk, NF output fragments of 16 columns (compile-time: the
// image is zero past the real width and af is zero past the sequence, so over-computing is exact and keeps the loops branch-free)
template <int NF>
__device__ __forceinline__ void frags_times_image(const uint4 (&af)[ATT_KF / 2], const unsigned char* vimg, int lane, f32x4_t (&o)[ATT_DC / 16]) {
#pragma unroll
  for (int n = 0; n < ATT_DC / 16; ++n) o[n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < ATT_KF / 2; ++kk) {
    Frag<bf16_t> a, bfr[NF];
    a.v = af[kk];
#pragma unroll
    for (int n = 0; n < NF; ++n) vimg_frag(vimg, kk, n, lane, bfr[n]);
#pragma unroll
    for (int n = 0; n < NF; ++n) frag_mma(a, bfr[n], o[n]);
  }
}
template <int NKS> struct AttnNF { static constexpr int value = NKS <= 1 ? 2 : (NKS <= 2 ? 4 : ATT_DC / 16); };

template <int NKS, bool WIN = false>
__global__ __launch_bounds__(256) void attn_fwd_kernel(tfpp_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;                       // 2 x 20 KB (phase 1), later the per-wave output strips
  unsigned char* vimg = smem + 2 * K_SLICE_BYTES;   // 80 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, kg = lane >> 4;
  const int T = p.T, d = p.d, nkf = WIN ? (T + 15) >> 4 : T >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
  const int q0 = blockIdx.x * ATT_TQ + wave * 16;
  const bf16_t* qh = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * T * p.ld_q + (size_t)h * d;
  const bf16_t* kh = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const bf16_t* vh = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  bf16_t* oh = reinterpret_cast<bf16_t*>(p.o) + (size_t)b * T * p.ld_o + (size_t)h * d;

  // phase trace (tools/attn_micro.py --trace): the forward does not use p.delta; when given, workgroup (0, 0) stamps its phases there
  const bool trace = p.delta != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0;
  const unsigned long long t0 = trace ? wall_clock64() : 0ull;
#define ATT_STAMP(k) do { if (trace) p.delta[k] = (float)(wall_clock64() - t0); } while (0)
  f32x4_t S[ATT_KF];
  const int qmine = WIN ? ((q0 + p16 < T) ? q0 + p16 : T - 1) : q0 + p16;  // windows: queries past T repeat the last one (never stored)
  attn_scores<NKS>(qh, p.ld_q, kh, p.ld_kv, T, d, qmine, kimg, tid, lane, S);
  ATT_STAMP(0);
  float lse = 0.f;
  if constexpr (WIN) {
    const float* brow = p.bias + ((size_t)h * T + qmine) * p.ld_b;
    const float* mrow = p.mask ? p.mask + ((size_t)(b % p.n_mask) * T + qmine) * p.ld_b : nullptr;
    attn_softmax_window(S, nkf, T, kg, p.scale, brow, mrow);
    if (p.p_out && q0 + p16 < T) {  // probabilities for the backward: 4 consecutive keys = one 8-byte store
      bf16_t* prow = reinterpret_cast<bf16_t*>(p.p_out) + ((size_t)bh * T + q0 + p16) * p.ld_p;
#pragma unroll
      for (int f = 0; f < ATT_KF; ++f)
        if (f < nkf) {
          const int key0 = f * 16 + kg * 4;
          if (key0 + 3 < T) {
            *reinterpret_cast<uint2*>(prow + key0) = make_uint2(pack_bf16x2(S[f][0], S[f][1]), pack_bf16x2(S[f][2], S[f][3]));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + r < T) prow[key0 + r] = f2bf(S[f][r]);
          }
        }
    }
  } else {
    lse = attn_softmax(S, nkf, p.scale);
  }
  ATT_STAMP(1);
  const size_t row = (size_t)bh * T + q0 + p16;  // flat (batch, head, query) index: dropout counter and lse / debug rows
  if (!WIN && kg == 0 && p.lse) p.lse[row] = lse;

  // dropout (same element index as the unfused path: row * T + key) and bf16 A fragments of P
  unsigned long long seed = p.seed;
  if (p.seed_offset) seed += *p.seed_offset * 0x9E3779B97F4A7C15ull;
  const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  uint4 pf[ATT_KF / 2];
#pragma unroll
  for (int kk = 0; kk < ATT_KF / 2; ++kk) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 2 * kk + (e >> 2), r = e & 3;
      float pv = 0.f;
      if (f < nkf) {
        pv = S[f][r];
        if (p.p_drop > 0.f) pv *= dropout_scale(seed, (unsigned long long)row * T + (f * 16 + kg * 4 + r), p.p_drop, inv_keep);
        if (p.debug_p) p.debug_p[row * T + (f * 16 + kg * 4 + r)] = pv;
      }
      v[e] = pv;
    }
    pf[kk] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }

  // O = P V, one 128-wide d chunk at a time
  unsigned char* strip = kimg + wave * O_STRIP_BYTES;
  for (int dc = 0; dc < d; dc += ATT_DC) {
    const int nd = (d - dc < ATT_DC) ? d - dc : ATT_DC, nfr = (nd + 15) >> 4;
    __syncthreads();  // previous chunk's readers are done with the V image (first pass: the K image is dead too)
    if (dc == 0) ATT_STAMP(2);
    stage_vimg(vh, p.ld_kv, T, d, dc, AttnNF<NKS>::value * 2, vimg, tid, WIN ? ATT_KF * 16 : 0);  // windows: the P.V loop walks all 320 image rows
    __syncthreads();
    if (dc == 0) ATT_STAMP(3);
    f32x4_t o[ATT_DC / 16];
    frags_times_image<AttnNF<NKS>::value>(pf, vimg, lane, o);
    if (dc == 0) ATT_STAMP(4);
    // C layout: o[n][r] = O[query kg*4 + r][d = dc + n*16 + p16] -> bf16 strip [16][128] -> 16-byte row segments
    bf16_t* st = reinterpret_cast<bf16_t*>(strip);
#pragma unroll
    for (int n = 0; n < ATT_DC / 16; ++n)
      if (n < nfr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(kg * 4 + r) * ATT_DC + n * 16 + p16] = f2bf(o[n][r]);
      }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the strip is wave-private, DS operations retire in order
    const int cpr = nfr * 2;
    for (int c = lane; c < 16 * cpr; c += 64) {
      const int r = c / cpr, ch = c - r * cpr, d0 = dc + ch * 8;
      if (d0 < d && (!WIN || q0 + r < T))
        *reinterpret_cast<uint4*>(oh + (size_t)(q0 + r) * p.ld_o + d0) = *reinterpret_cast<const uint4*>(st + r * ATT_DC + ch * 8);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// backward.  With Pd = P .* M (M = dropout mask / keep), O = Pd V:
//   dPd = dO V^T      delta[q] = sum_j dPd[q][j] Pd[q][j] = sum_d dO[q][d] O[q][d]      dS = scale * P .* (dPd .* M - delta)
//   dQ = dS K         dK = dS^T Q         dV = Pd^T dO
// P is recomputed from Q, K and the saved log-sum-exp (no softmax reduction, no stored probabilities).
//
// attn_bwd_dq_kernel   grid (T / 64, B * nh): as the forward -- a lane owns one query and 80 keys of S^T and dPd^T; dS^T fragments
//                      are the A operand of dS K exactly as P was of P V (K staged as the "V image").  Also writes delta[] for the
//                      second kernel.
// attn_bwd_dkv_kernel  grid (T / 64, B * nh) over KEY tiles: the products are taken unswapped (S = Q K^T: A = Q rows staged through
//                      LDS, B = this wave's 16 K rows), so a lane owns one KEY and 80 queries; Pd^T / dS^T fragments feed
//                      dV = Pd^T dO and dK = dS^T Q with dO / Q staged as the transposable image.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_strip_chunk(const f32x4_t (&o)[ATT_DC / 16], int nfr, unsigned char* strip, bf16_t* dst, long ld, int row0,
                                                  int dc, int d, int lane) {
  const int p16 = lane & 15, kg = lane >> 4;
  bf16_t* st = reinterpret_cast<bf16_t*>(strip);
#pragma unroll
  for (int n = 0; n < ATT_DC / 16; ++n)
    if (n < nfr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) st[(kg * 4 + r) * ATT_DC + n * 16 + p16] = f2bf(o[n][r]);
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  const int cpr = nfr * 2;
  for (int c = lane; c < 16 * cpr; c += 64) {
    const int r = c / cpr, ch = c - r * cpr, d0 = dc + ch * 8;
    if (d0 < d) *reinterpret_cast<uint4*>(dst + (size_t)(row0 + r) * ld + d0) = *reinterpret_cast<const uint4*>(st + r * ATT_DC + ch * 8);
  }
}

template <int NKS>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(tfpp_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + 2 * K_SLICE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, kg = lane >> 4;
  const int T = p.T, d = p.d, nkf = T >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
  const int q0 = blockIdx.x * ATT_TQ + wave * 16;
  const bf16_t* qh = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * T * p.ld_q + (size_t)h * d;
  const bf16_t* kh = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const bf16_t* vh = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const bf16_t* oh = reinterpret_cast<const bf16_t*>(p.o) + (size_t)b * T * p.ld_o + (size_t)h * d;
  const bf16_t* doh = reinterpret_cast<const bf16_t*>(p.d_o) + (size_t)b * T * p.ld_o + (size_t)h * d;
  bf16_t* dqh = reinterpret_cast<bf16_t*>(p.dq) + (size_t)b * T * p.ld_q + (size_t)h * d;
  const size_t row = (size_t)bh * T + q0 + p16;

  // delta of this lane's query: its d chunks (kg, kg + 4, ...) then the three partner lanes
  float delta = 0.f;
  for (int d0 = kg * 8; d0 < d; d0 += 32) {
    float a[8], c[8];
    load_vec<bf16_t>(doh + (size_t)(q0 + p16) * p.ld_o + d0, a);
    load_vec<bf16_t>(oh + (size_t)(q0 + p16) * p.ld_o + d0, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) delta += a[e] * c[e];
  }
  delta += __shfl_xor(delta, 16, 64);
  delta += __shfl_xor(delta, 32, 64);
  if (kg == 0) p.delta[row] = delta;
  const float lse = p.lse[row];

  f32x4_t S[ATT_KF];
  attn_scores<NKS>(qh, p.ld_q, kh, p.ld_kv, T, d, q0 + p16, kimg, tid, lane, S);
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f)
#pragma unroll
    for (int r = 0; r < 4; ++r) S[f][r] = (f < nkf) ? __expf(S[f][r] * p.scale - lse) : 0.f;  // P
  __syncthreads();  // every wave is done with the K slices before they are overwritten with V slices
  f32x4_t dP[ATT_KF];
  attn_scores<NKS>(doh, p.ld_o, vh, p.ld_kv, T, d, q0 + p16, kimg, tid, lane, dP);  // dPd^T = V dO^T

  unsigned long long seed = p.seed;
  if (p.seed_offset) seed += *p.seed_offset * 0x9E3779B97F4A7C15ull;
  const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  uint4 dsf[ATT_KF / 2];
#pragma unroll
  for (int kk = 0; kk < ATT_KF / 2; ++kk) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 2 * kk + (e >> 2), r = e & 3;
      float g = 0.f;
      if (f < nkf) {
        g = dP[f][r];
        if (p.p_drop > 0.f) g *= dropout_scale(seed, (unsigned long long)row * T + (f * 16 + kg * 4 + r), p.p_drop, inv_keep);
        g = p.scale * S[f][r] * (g - delta);
      }
      v[e] = g;
    }
    dsf[kk] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }

  unsigned char* strip = kimg + wave * O_STRIP_BYTES;
  for (int dc = 0; dc < d; dc += ATT_DC) {
    const int nd = (d - dc < ATT_DC) ? d - dc : ATT_DC, nfr = (nd + 15) >> 4;
    __syncthreads();
    stage_vimg(kh, p.ld_kv, T, d, dc, AttnNF<NKS>::value * 2, vimg, tid);  // dQ = dS K: K as the transposable image
    __syncthreads();
    f32x4_t o[ATT_DC / 16];
    frags_times_image<AttnNF<NKS>::value>(dsf, vimg, lane, o);
    store_strip_chunk(o, nfr, strip, dqh, p.ld_q, q0, dc, d, lane);
  }
}

template <int NKS>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(tfpp_attn_params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* kimg = smem;
  unsigned char* vimg = smem + 2 * K_SLICE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, kg = lane >> 4;
  const int T = p.T, d = p.d, nkf = T >> 4;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
  const int k0 = blockIdx.x * ATT_TQ + wave * 16;  // this wave's 16 keys
  const bf16_t* qh = reinterpret_cast<const bf16_t*>(p.q) + (size_t)b * T * p.ld_q + (size_t)h * d;
  const bf16_t* kh = reinterpret_cast<const bf16_t*>(p.k) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const bf16_t* vh = reinterpret_cast<const bf16_t*>(p.v) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const bf16_t* doh = reinterpret_cast<const bf16_t*>(p.d_o) + (size_t)b * T * p.ld_o + (size_t)h * d;
  bf16_t* dkh = reinterpret_cast<bf16_t*>(p.dk) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  bf16_t* dvh = reinterpret_cast<bf16_t*>(p.dv) + (size_t)b * T * p.ld_kv + (size_t)h * d;
  const size_t rbase = (size_t)bh * T;

  // unswapped scores: "k operand" (staged, fragment rows) = Q, "q operand" (one row per lane) = this lane's key
  f32x4_t S[ATT_KF];
  attn_scores<NKS>(kh, p.ld_kv, qh, p.ld_q, T, d, k0 + p16, kimg, tid, lane, S);  // S[f][r]: query f*16 + kg*4 + r, key k0 + p16
#pragma unroll
  for (int f = 0; f < ATT_KF; ++f) {
    if (f < nkf) {
      const float4 l4 = *reinterpret_cast<const float4*>(p.lse + rbase + f * 16 + kg * 4);
      S[f][0] = __expf(S[f][0] * p.scale - l4.x); S[f][1] = __expf(S[f][1] * p.scale - l4.y);
      S[f][2] = __expf(S[f][2] * p.scale - l4.z); S[f][3] = __expf(S[f][3] * p.scale - l4.w);
    } else {
      S[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  f32x4_t dP[ATT_KF];
  attn_scores<NKS>(vh, p.ld_kv, doh, p.ld_o, T, d, k0 + p16, kimg, tid, lane, dP);  // dPd[query][key] = dO V^T

  unsigned long long seed = p.seed;
  if (p.seed_offset) seed += *p.seed_offset * 0x9E3779B97F4A7C15ull;
  const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
  uint4 pdf[ATT_KF / 2], dsf[ATT_KF / 2];  // A fragments of Pd^T and dS^T: row = key (lane & 15), contraction = queries
#pragma unroll
  for (int kk = 0; kk < ATT_KF / 2; ++kk) {
    float pv[8], gv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = 2 * kk + (e >> 2), r = e & 3;
      pv[e] = 0.f; gv[e] = 0.f;
      if (f < nkf) {
        const int qrow = f * 16 + kg * 4 + r;
        float m = 1.f;
        if (p.p_drop > 0.f) m = dropout_scale(seed, (unsigned long long)(rbase + qrow) * T + (k0 + p16), p.p_drop, inv_keep);
        pv[e] = S[f][r] * m;
        gv[e] = p.scale * S[f][r] * (dP[f][r] * m - p.delta[rbase + qrow]);
      }
    }
    pdf[kk] = make_uint4(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]), pack_bf16x2(pv[4], pv[5]), pack_bf16x2(pv[6], pv[7]));
    dsf[kk] = make_uint4(pack_bf16x2(gv[0], gv[1]), pack_bf16x2(gv[2], gv[3]), pack_bf16x2(gv[4], gv[5]), pack_bf16x2(gv[6], gv[7]));
  }

  unsigned char* strip = kimg + wave * O_STRIP_BYTES;
  for (int dc = 0; dc < d; dc += ATT_DC) {
    const int nd = (d - dc < ATT_DC) ? d - dc : ATT_DC, nfr = (nd + 15) >> 4;
    f32x4_t o[ATT_DC / 16];
    __syncthreads();
    stage_vimg(doh, p.ld_o, T, d, dc, AttnNF<NKS>::value * 2, vimg, tid);  // dV = Pd^T dO
    __syncthreads();
    frags_times_image<AttnNF<NKS>::value>(pdf, vimg, lane, o);
    store_strip_chunk(o, nfr, strip, dvh, p.ld_kv, k0, dc, d, lane);
    __syncthreads();
    stage_vimg(qh, p.ld_q, T, d, dc, AttnNF<NKS>::value * 2, vimg, tid);   // dK = dS^T Q
    __syncthreads();
    frags_times_image<AttnNF<NKS>::value>(dsf, vimg, lane, o);
    store_strip_chunk(o, nfr, strip, dkh, p.ld_kv, k0, dc, d, lane);
  }
}
}  // namespace

static bool attn_shape_ok(const tfpp_attn_params& p, int dtype) {
  if (dtype != TFPP_BF16 || !p.q || !p.k || !p.v || !p.o) return false;
  if (p.T < ATT_TQ || p.T % ATT_TQ != 0 || p.T > ATT_KF * 16) return false;
  if (p.d < 8 || p.d % 8 != 0 || p.d > 384) return false;
  if ((p.ld_q | p.ld_kv | p.ld_o) % 8 != 0) return false;
  if (((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.o) & 15) return false;
  return p.B >= 1 && p.nh >= 1 && p.p_drop >= 0.f && p.p_drop < 1.f;
}

extern "C" int tfpp_attn_supported(const tfpp_attn_params* p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_FUSED_ATTN"); return (e && e[0] == '0') ? 0 : 1; }();
  return (p && on && attn_shape_ok(*p, dtype)) ? 1 : 0;
}

template <int NKS> static int launch_attn_fwd(const tfpp_attn_params& p, hipStream_t st) {
  constexpr size_t lds = 2 * K_SLICE_BYTES + V_IMG_BYTES;
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<NKS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(attn_fwd_kernel<NKS>, dim3(p.T / ATT_TQ, p.B * p.nh), dim3(256), lds, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_attn_fwd(const tfpp_attn_params* p, int dtype, void* stream) {
  if (!p || !attn_shape_ok(*p, dtype)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nks = (p->d + 31) / 32;
  if (nks <= 1) return launch_attn_fwd<1>(*p, st);
  if (nks <= 2) return launch_attn_fwd<2>(*p, st);
  if (nks <= 5) return launch_attn_fwd<5>(*p, st);
  return launch_attn_fwd<12>(*p, st);
}

// ---- window attention (Video-Swin): any T <= 320, d <= 32 in this model (one k-step), dense bias / mask, probabilities saved on request
extern "C" int tfpp_attn_window_fwd(const tfpp_attn_params* p, int dtype, void* stream) {
  if (!p || dtype != TFPP_BF16 || !p->q || !p->k || !p->v || !p->o || !p->bias) return TFPP_EINVAL;
  if (p->T < 1 || p->T > ATT_KF * 16 || p->d < 8 || p->d % 8 != 0 || p->d > 64 || p->p_drop != 0.f) return TFPP_EINVAL;
  if ((p->ld_q | p->ld_kv | p->ld_o) % 8 != 0 || p->ld_b % 4 != 0 || p->ld_b < ((p->T + 15) / 16) * 16 || (p->mask && p->n_mask < 1)) return TFPP_EINVAL;
  if (p->p_out && (p->ld_p % 4 != 0 || p->ld_p < p->T)) return TFPP_EINVAL;
  if (((uintptr_t)p->q | (uintptr_t)p->k | (uintptr_t)p->v | (uintptr_t)p->o | (uintptr_t)p->bias | (uintptr_t)p->mask) & 15) return TFPP_EINVAL;
  constexpr size_t lds = 2 * K_SLICE_BYTES + V_IMG_BYTES;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((p->T + ATT_TQ - 1) / ATT_TQ, p->B * p->nh);
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (p->d <= 32) hipLaunchKernelGGL((attn_fwd_kernel<1, true>), grid, dim3(256), lds, st, *p);
  else hipLaunchKernelGGL((attn_fwd_kernel<2, true>), grid, dim3(256), lds, st, *p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

__global__ void window_bias_dense_kernel(const float* __restrict__ table, const int* __restrict__ rel_index, float* __restrict__ dense, int heads,
                                         int n, long ld_b) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)heads * n * ld_b) return;
  const int j = (int)(t % ld_b);
  const int i = (int)((t / ld_b) % n);
  const int h = (int)(t / (ld_b * n));
  dense[t] = j < n ? table[(size_t)rel_index[(size_t)i * n + j] * heads + h] : 0.f;
}

extern "C" int tfpp_window_bias_dense(const float* table, const int32_t* rel_index, float* dense, int heads, int n, int64_t ld_b, void* stream) {
  if (!table || !rel_index || !dense || heads < 1 || n < 1 || ld_b < n) return TFPP_EINVAL;
  const long total = (long)heads * n * ld_b;
  hipLaunchKernelGGL(window_bias_dense_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, (const int*)rel_index,
                     dense, heads, n, (long)ld_b);
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <int NKS> static int launch_attn_bwd(const tfpp_attn_params& p, hipStream_t st) {
  constexpr size_t lds = 2 * K_SLICE_BYTES + V_IMG_BYTES;
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<NKS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<NKS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const dim3 grid(p.T / ATT_TQ, p.B * p.nh);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<NKS>, grid, dim3(256), lds, st, p);   // also writes delta[]
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<NKS>, grid, dim3(256), lds, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_attn_bwd(const tfpp_attn_params* p, int dtype, void* stream) {
  if (!p || !attn_shape_ok(*p, dtype) || !p->d_o || !p->dq || !p->dk || !p->dv || !p->lse || !p->delta) return TFPP_EINVAL;
  if (((uintptr_t)p->d_o | (uintptr_t)p->dq | (uintptr_t)p->dk | (uintptr_t)p->dv) & 15) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nks = (p->d + 31) / 32;
  if (nks <= 1) return launch_attn_bwd<1>(*p, st);
  if (nks <= 2) return launch_attn_bwd<2>(*p, st);
  if (nks <= 5) return launch_attn_bwd<5>(*p, st);
  return launch_attn_bwd<12>(*p, st);
}
