// Barrier-free implicit-GEMM convolution / linear layer (forward and data gradient): MFMA fragments are loaded
// straight from HBM/L2 into registers.
//
// Why this works on gfx950: for v_mfma_f32_16x16x32_bf16 lane l of the A (resp. B) fragment needs 8 consecutive K elements
// of row (l&15) at K offset (l>>4)*8 -- for a K-contiguous operand (NHWC activations: channels of one (r,s) tap;
// packed weights [n][(r,s,c)]) that is ONE aligned 16-byte load per lane.  So neither operand has to be staged or
// transposed through LDS, a workgroup needs no barrier, and every wave is an independent stream of
// {8 x 16-byte loads -> 16 MFMAs} with the next K-step's fragments in flight while the current one multiplies.
// Operand reuse between the four waves of a workgroup (2x2 wave tiles share A rows / B columns) is served by the
// CU's vector L1 and the XCD L2 instead of LDS.  The LDS-staged kernel (gemm_kernels.hip) measured 60-160 TFLOP/s on
// this model's shapes because its load->LDS->barrier->MFMA->barrier loop exposes one memory latency per 32-deep K tile.
#include "gemm_core.cuh"
#include "gemm_internal.h"

// A fragment is 8 consecutive K elements per lane = PARTS 16-byte vectors (bf16: 1, fp32: 2).  Each part is bounds- and
// tap-checked on its own, so fp32 operands only need K / channel counts that are multiples of 4.
template <typename T> struct FragParts;
template <> struct FragParts<bf16_t> {
  static constexpr int PARTS = 1;
  __device__ static __forceinline__ void set(Frag<bf16_t>& f, int, const uint4& u) { f.v = u; }
};
template <> struct FragParts<float> {
  static constexpr int PARTS = 2;
  __device__ static __forceinline__ void set(Frag<float>& f, int h, const uint4& u) {
    f.v[4 * h + 0] = __uint_as_float(u.x); f.v[4 * h + 1] = __uint_as_float(u.y);
    f.v[4 * h + 2] = __uint_as_float(u.z); f.v[4 * h + 3] = __uint_as_float(u.w);
  }
};

// wave tile (FM*16) x (FN*16); workgroup = 2x2 waves
template <typename T, int FM, int FN>
__global__ __launch_bounds__(256) void conv_gemm_direct_kernel(tfpp_conv_params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kg = (lane >> 4) * 8;
  const int g = blockIdx.z;
  const int bm0 = (blockIdx.x * 2 + (wave >> 1)) * (FM * 16);
  const int bn0 = (blockIdx.y * 2 + (wave & 1)) * (FN * 16);
  const int M = p.B * p.Hd * p.Wd, K = p.R * p.S * p.ks_g;
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src) + g * p.ks_g;
  const T* __restrict__ wk = reinterpret_cast<const T*>(p.w) + (size_t)g * p.n_g * K;
  if (bm0 >= M || bn0 >= p.n_g) return;  // wave-uniform: nothing to do for this wave tile

  int a_b[FM], a_h0[FM], a_w0[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = bm0 + i * 16 + r16;
    if (m < M) {
      const int hw = p.Hd * p.Wd, b = m / hw, pix = m - b * hw, hd = pix / p.Wd, wd = pix - hd * p.Wd;
      a_b[i] = b;
      if (p.mode == 0) { a_h0[i] = hd * p.stride - p.pad; a_w0[i] = wd * p.stride - p.pad; }
      else { a_h0[i] = hd + p.pad; a_w0[i] = wd + p.pad; }
    } else {
      a_b[i] = -1; a_h0[i] = 0; a_w0[i] = 0;
    }
  }
  const T* b_row[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = bn0 + j * 16 + r16;
    b_row[j] = (n < p.n_g) ? wk + (size_t)n * K : nullptr;
  }
  const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);
  const T* a_row[FM];  // pointwise fast path: the source pixel never changes along K
#pragma unroll
  for (int i = 0; i < FM; ++i)
    a_row[i] = (pointwise && a_b[i] >= 0) ? src + ((size_t)(a_b[i] * p.Hs + a_h0[i]) * p.Ws + a_w0[i]) * p.src_ld : nullptr;

  constexpr int VEC = ElemTraits<T>::VEC, PARTS = FragParts<T>::PARTS;
  auto load = [&](int ks, Frag<T> (&fa)[FM], Frag<T> (&fb)[FN]) {
#pragma unroll
    for (int h = 0; h < PARTS; ++h) {
      const int k0 = ks * 32 + kg + h * VEC;
      const bool kin = k0 < K;
      if (pointwise) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          uint4 u = make_uint4(0, 0, 0, 0);
          if (kin && a_row[i]) u = *reinterpret_cast<const uint4*>(a_row[i] + k0);
          FragParts<T>::set(fa[i], h, u);
        }
      } else {
        const int rs = k0 / p.ks_g, c = k0 - rs * p.ks_g, r = rs / p.S, s = rs - r * p.S;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
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
          uint4 u = make_uint4(0, 0, 0, 0);
          if (kin && ok && a_b[i] >= 0) u = *reinterpret_cast<const uint4*>(src + ((size_t)(a_b[i] * p.Hs + hs) * p.Ws + ws) * p.src_ld + c);
          FragParts<T>::set(fa[i], h, u);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (kin && b_row[j]) u = *reinterpret_cast<const uint4*>(b_row[j] + k0);
        FragParts<T>::set(fb[j], h, u);
      }
    }
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  auto mma = [&](Frag<T> (&fa)[FM], Frag<T> (&fb)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) frag_mma(fa[i], fb[j], acc[i][j]);
  };

  // two named fragment sets (static indexing keeps them in registers): loads of step ks+1 overlap the MFMAs of step ks
  const int nks = (K + 31) / 32;
  Frag<T> a0[FM], b0[FN], a1[FM], b1[FN];
  load(0, a0, b0);
  for (int ks = 0; ks < nks; ks += 2) {
    if (ks + 1 < nks) load(ks + 1, a1, b1);
    mma(a0, b0);
    if (ks + 2 < nks) load(ks + 2, a0, b0);
    if (ks + 1 < nks) mma(a1, b1);
  }

  // epilogue (same semantics as the LDS kernel)
  const int hw = p.Hd * p.Wd;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm0 + i * 16 + (lane >> 4) * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = bn0 + j * 16 + r16;
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

template <typename T, int FM, int FN> static int launch_direct(const tfpp_conv_params& p, hipStream_t st) {
  const long M = (long)p.B * p.Hd * p.Wd;
  dim3 grid(cdiv(M, 2 * FM * 16), cdiv(p.n_g, 2 * FN * 16), p.G);
  hipLaunchKernelGGL((conv_gemm_direct_kernel<T, FM, FN>), grid, dim3(256), 0, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// variant code = 100 + FM*10 + FN  (FM in {2,4}; FN in {1,2,3,4}); wave tile (16 FM) x (16 FN), workgroup tile twice that
int conv_direct_variant(const tfpp_conv_params& p, int dtype) {
  const long M = (long)p.B * p.Hd * p.Wd;
  const int N = p.n_g;
  if (dtype == TFPP_F32) return 100 + 20 + (N <= 32 ? 1 : 2);  // fp32 fragments are 8 VGPRs: keep the wave tile at 32 x 32
  const int fn = N <= 32 ? 1 : (N <= 64 ? 2 : (N <= 96 ? 3 : 4));
  const long blocks4 = (long)cdiv(M, 128) * cdiv(N, 32 * fn) * p.G;
  const int fm = blocks4 < 512 ? 2 : 4;  // small grids: halve the wave tile to double the number of waves
  return 100 + fm * 10 + fn;
}

int conv_gemm_direct(const tfpp_conv_params& p, int dtype, hipStream_t st) {
  const int v = conv_direct_variant(p, dtype) - 100, fm = v / 10, fn = v % 10;
#define DIRECT_CASE(TT, A, B) if (fm == A && fn == B) return launch_direct<TT, A, B>(p, st)
  if (dtype == TFPP_F32) {
    DIRECT_CASE(float, 2, 1);
    DIRECT_CASE(float, 2, 2);
    return TFPP_EINVAL;
  }
  DIRECT_CASE(bf16_t, 2, 1); DIRECT_CASE(bf16_t, 2, 2); DIRECT_CASE(bf16_t, 2, 3); DIRECT_CASE(bf16_t, 2, 4);
  DIRECT_CASE(bf16_t, 4, 1); DIRECT_CASE(bf16_t, 4, 2); DIRECT_CASE(bf16_t, 4, 3); DIRECT_CASE(bf16_t, 4, 4);
#undef DIRECT_CASE
  return TFPP_EINVAL;
}
