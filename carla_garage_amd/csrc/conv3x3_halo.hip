// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) with the input tile staged ONCE in LDS, bf16.
//
// The implicit-GEMM kernels gather the A operand per K-tile, so a 3x3 layer pulls every input pixel through L2 -> CU nine
// times; on the narrow 3x3 layers of this model (full-resolution perspective decoders: 3.1 M pixels x 32 channels;
// RegNet grouped convs: 24 channels per group) that L2 traffic, not MFMA, is what bounds them.  Here a workgroup
// owns an 8 x 32 pixel output tile of one group: it loads the (8+2) x (32+2) x Cin_g halo once (contiguous image
// [halo pixel][Cin_g]) and the group's weights [n][K] once, and then every tap is just an LDS address offset:
//   MFMA K index kk = tap * Cin_g + c  ->  lane (pixel p = l & 15, k-group g = l >> 4) reads the 16-byte chunk
//   halo[(row + 1 + dr(tap)) * 34 + col + 1 + dc(tap)][c0 .. c0+7],   tap = (32 j + 8 g) / Cin_g,  c0 = (32 j + 8 g) % Cin_g.
// With Cin_g = 32 the 64 lanes of a fragment read 1 KB of contiguous LDS (conflict-free); weight rows are skewed by 16 B.
// One barrier per workgroup (after staging); 4 waves x (2 rows x 32 pixels) x FN 16-channel fragments.
// Data gradient (mode 1) at stride 1 is the same gather with the tap offsets negated (the caller passes the transposed pack).
// GEO 1 / 2 (round 2): the stride-2 3x3 convs that open every RegNet stage.  Forward: the 8 x 32 OUTPUT tile reads a 17 x 65 input halo
// and the pixel -> LDS address map is scaled by 2.  Data gradient: dx = stride-1 correlation of the ZERO-STUFFED dy (Z[2a][2b] = dy[a][b])
// with the same transposed pack, so only the loader changes (3 of 4 halo pixels are zeros; the layers are HBM-bound, the idle MFMA
// work is free).  The implicit GEMM ran these at 119 / 242 us (stage 1, bs = 12) against a 28 us HBM bound.
#include "gemm_core.h"
#include "gemm_internal.h"
#include "bn_rows.h"
#include <cstdlib>

namespace {
constexpr int TH = 8, TW = 32;
template <int GEO> struct HaloGeo { static constexpr int HH = GEO == 1 ? 2 * TH + 1 : TH + 2, HWID = GEO == 1 ? 2 * TW + 1 : TW + 2; };

// CV = Cin_g / 8 at compile time (0: run-time loop): with CV known the staging loops are fully unrolled, so a thread issues all of
// its ~11 global loads before the first LDS store instead of paying one memory latency per 16-byte chunk.
// INBN (round 6): the source exists only as (raw output of the 1x1 convolution in front, BatchNorm statistics): every staged chunk becomes
// relu(raw * scale[c] + shift[c]) on its way into LDS (out-of-image chunks stay zero: the padding is applied AFTER the activation), and when
// in_bn.partial is set the workgroup first adds the statistics rows of its group's input channels (bn_rows.h) -- beside its tile loads, which
// are already in flight.  The normalised tensor is never written (tfpp.h).
template <int FN, int CV, bool BNS = false, int GEO = 0, bool INBN = false>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(tfpp_conv_params p, int tiles_w, int tiles_h, int ksteps_rt) {
  typedef bf16_t T;
  constexpr int FM = 4, HH = HaloGeo<GEO>::HH, HWID = HaloGeo<GEO>::HWID;
  static_assert(!(BNS && GEO != 0), "the statistics epilogue exists for the stride-1 geometry only");
  static_assert(!(INBN && (BNS || GEO == 2)), "normalise-on-load is a forward feature");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cin = CV ? CV * 8 : p.ks_g, cv = cin >> 3, K = 9 * cin;
  const int ksteps = CV ? (9 * CV * 8 + 31) / 32 : ksteps_rt;
  const int g = blockIdx.y;
  int t = blockIdx.x;
  const int tw = t % tiles_w; t /= tiles_w;
  const int th = t % tiles_h;
  const int b = t / tiles_h;
  const int h0 = th * TH, w0 = tw * TW;
  const int H = p.Hd, W = p.Wd;
  const int kpitch = ksteps * 32 + 8;  // weight row pitch in elements (16-byte skew: conflict-free fragment reads)
  T* halo = reinterpret_cast<T*>(smem);
  T* wl = halo + HH * HWID * cin;

  // ---- stage the halo tile (zero outside the image) and the group's weights (zero rows / K tail)
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src) + g * cin;
  const T* __restrict__ wk = reinterpret_cast<const T*>(p.w) + (size_t)g * p.n_g * K;
  const int kc_row = kpitch >> 3;  // 16-byte chunks per LDS weight row
  auto halo_chunk = [&](int q, bool& inside) {
    const int pix = q / cv, c = q - pix * cv;
    const int hr = pix / HWID, hc = pix - hr * HWID;
    uint4 v = make_uint4(0, 0, 0, 0);
    inside = false;
    if constexpr (GEO == 0) {
      const int h = h0 + hr - 1, w = w0 + hc - 1;
      inside = h >= 0 && h < H && w >= 0 && w < W;
      if (inside) v = *reinterpret_cast<const uint4*>(src + ((size_t)(b * H + h) * W + w) * p.src_ld + c * 8);
    } else if constexpr (GEO == 1) {  // forward, stride 2: input pixel (2 h0 - 1 + hr, 2 w0 - 1 + hc) of the Hs x Ws source
      const int h = 2 * h0 + hr - 1, w = 2 * w0 + hc - 1;
      inside = h >= 0 && h < p.Hs && w >= 0 && w < p.Ws;
      if (inside) v = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.Hs + h) * p.Ws + w) * p.src_ld + c * 8);
    } else {  // data gradient, stride 2: the zero-stuffed gradient Z[2a][2b] = dy[a][b]
      const int hz = h0 + hr - 1, wz = w0 + hc - 1;
      inside = hz >= 0 && wz >= 0 && !((hz | wz) & 1) && (hz >> 1) < p.Hs && (wz >> 1) < p.Ws;
      if (inside) v = *reinterpret_cast<const uint4*>(src + ((size_t)(b * p.Hs + (hz >> 1)) * p.Ws + (wz >> 1)) * p.src_ld + c * 8);
    }
    return v;
  };
  // normalise-on-load: scale / shift of this group's input channels in LDS; the transform of one staged chunk
  __shared__ float in_sc[INBN ? 64 : 1], in_sh[INBN ? 64 : 1];
  auto in_bn_chunk = [&](uint4 v, int q) {
    const int c8 = (q % cv) * 8;
    float f[8];
    unpack16<T>(v, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t = f[e] * in_sc[c8 + e] + in_sh[c8 + e];
      f[e] = p.in_relu ? (t > 0.f ? t : 0.f) : t;
    }
    return pack16<T>(f);
  };
  auto weight_chunk = [&](int q) {
    const int n = q / kc_row, kc = q - n * kc_row;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < p.n_g && kc * 8 < K) v = *reinterpret_cast<const uint4*>(wk + (size_t)n * K + kc * 8);
    return v;
  };
  if constexpr (CV > 0) {
    constexpr int HCH = HH * HWID * CV, HIT = (HCH + 255) / 256;
    constexpr int WCH = FN * 16 * (((9 * CV * 8 + 31) / 32) * 4 + 1), WIT = (WCH + 255) / 256;
    if constexpr (INBN) {  // the statistics of this group's (<= 64) input channels FIRST: its registers are dead before the tile loads are
      // issued -- one more memory round trip than overlapping the two, but the kernel keeps 4 workgroups per CU (overlapped: 196 registers, 2)
      __shared__ double in_sm[512];
      bn_block_scale_shift<12>(p.in_bn, g * cin, cin, blockIdx.x == 0, blockIdx.x == 0 && blockIdx.y == 0, in_sm, in_sc, in_sh);
    }
    uint4 hv[HIT], wv[WIT];
    unsigned inside_bits = 0u;
    static_assert(HIT <= 32, "one validity bit per staged chunk");
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int q = tid + it * 256;
      bool inside = false;
      hv[it] = q < HCH ? halo_chunk(q, inside) : make_uint4(0, 0, 0, 0);
      if (INBN && inside) inside_bits |= 1u << it;
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) { const int q = tid + it * 256; wv[it] = q < WCH ? weight_chunk(q) : make_uint4(0, 0, 0, 0); }
    if constexpr (INBN) {
#pragma unroll
      for (int it = 0; it < HIT; ++it)
        if (inside_bits & (1u << it)) hv[it] = in_bn_chunk(hv[it], tid + it * 256);
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) { const int q = tid + it * 256; if (q < HCH) *reinterpret_cast<uint4*>(halo + (size_t)q * 8) = hv[it]; }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int q = tid + it * 256;
      if (q < WCH) { const int n = q / kc_row, kc = q - n * kc_row; *reinterpret_cast<uint4*>(wl + (size_t)n * kpitch + kc * 8) = wv[it]; }
    }
  } else {
    if constexpr (INBN) {
      __shared__ double in_sm[512];
      bn_block_scale_shift<12>(p.in_bn, g * cin, cin, blockIdx.x == 0, blockIdx.x == 0 && blockIdx.y == 0, in_sm, in_sc, in_sh);
    }
    for (int q = tid; q < HH * HWID * cv; q += 256) {
      bool inside = false;
      uint4 v = halo_chunk(q, inside);
      if (INBN && inside) v = in_bn_chunk(v, q);
      *reinterpret_cast<uint4*>(halo + (size_t)q * 8) = v;
    }
    for (int q = tid; q < FN * 16 * kc_row; q += 256) {
      const int n = q / kc_row, kc = q - n * kc_row;
      *reinterpret_cast<uint4*>(wl + (size_t)n * kpitch + kc * 8) = weight_chunk(q);
    }
  }
  __syncthreads();

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // this wave: tile rows 2*wave, 2*wave+1; fragment i covers row 2*wave + (i >> 1), columns (i & 1) * 16 + (lane & 15)
  const int p16 = lane & 15, kg = lane >> 4;
  int a_pix[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i)
    a_pix[i] = GEO == 1 ? 2 * (2 * wave + (i >> 1)) * HWID + 2 * ((i & 1) * 16 + p16) + HWID + 1  // centre tap of output pixel (row, col)
                        : (2 * wave + (i >> 1) + 1) * HWID + (i & 1) * 16 + p16 + 1;
  const int sgn = p.mode == 0 ? 1 : -1;
  for (int j = 0; j < ksteps; ++j) {
    const int kk0 = j * 32 + kg * 8;
    const int tap = kk0 / cin, c0 = kk0 - tap * cin;
    const int r = tap / 3, s = tap - r * 3;
    const int off = sgn * ((r - 1) * HWID + (s - 1));
    Frag<T> fa[FM], fb[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      fa[i].v = make_uint4(0, 0, 0, 0);
      if (tap < 9) fa[i].v = *reinterpret_cast<const uint4*>(halo + (size_t)(a_pix[i] + off) * cin + c0);
    }
#pragma unroll
    for (int n = 0; n < FN; ++n) fb[n].v = *reinterpret_cast<const uint4*>(wl + (size_t)(n * 16 + p16) * kpitch + kk0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int n = 0; n < FN; ++n) frag_mma(fa[i], fb[n], acc[i][n]);
  }

  // ---- fused BatchNorm statistics (same contract as conv_gemm_kernel: rows pre-zeroed, fp32 atomics per workgroup)
  if (p.stats_partial) {
    __syncthreads();  // staging buffers are dead
    float* st = reinterpret_cast<float*>(smem);  // [2][4 waves][FN][16]
#pragma unroll
    for (int n = 0; n < FN; ++n) {
      float sum = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int h = h0 + 2 * wave + (i >> 1), w = w0 + (i & 1) * 16 + kg * 4 + r;
          if (h < H && w < W) { const float v = acc[i][n][r] * p.alpha; sum += v; sq += v * v; }
        }
      sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
      if (lane < 16) { st[((0 * 4 + wave) * FN + n) * 16 + lane] = sum; st[((1 * 4 + wave) * FN + n) * 16 + lane] = sq; }
    }
    __syncthreads();
    if (wave == 0 && lane < 16) {
      const int ctot = p.G * p.n_g;
#pragma unroll
      for (int n = 0; n < FN; ++n) {
        const int ch = n * 16 + lane;
        if (ch < p.n_g) {
          float sum = 0.f, sq = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < 4; ++w2) { sum += st[((0 * 4 + w2) * FN + n) * 16 + lane]; sq += st[((1 * 4 + w2) * FN + n) * 16 + lane]; }
          float* row = p.stats_partial + (size_t)(blockIdx.x % p.stats_rows) * 2 * ctot;
          if (p.stats_store) { row[g * p.n_g + ch] = sum; row[ctot + g * p.n_g + ch] = sq; }  // one writer per cell: nothing to zero (tfpp.h)
          else { atomicAdd(row + g * p.n_g + ch, sum); atomicAdd(row + ctot + g * p.n_g + ch, sq); }
        }
      }
    }
  }

  // ---- epilogue: C fragment row (kg * 4 + r) is the pixel, column p16 the channel
  if (epi_vec_ok(p)) {  // coalesced 16-pixel passes through a per-wave LDS strip (staging buffers are dead)
    __syncthreads();
    float* strip = reinterpret_cast<float*>(smem) + wave * EpiStrip<FN>::FLOATS;
    if constexpr (BNS) {  // fused BatchNorm-backward statistics (tfpp.h): one row of bns_partial per 8 x 32 pixel tile; own instantiation
      BnsAcc<FN, FM> bns;
      bns.init(p, lane, 0, g);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int h = h0 + 2 * wave + (i >> 1), wc = w0 + (i & 1) * 16;
        bns.prefetch(p, lane, i, (long)(b * H + h) * W + wc, h < H ? W - wc : 0, 0, g);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int h = h0 + 2 * wave + (i >> 1), wc = w0 + (i & 1) * 16;
        epi_pass_bf16<FN, FM, true>(p, acc[i], strip, lane, (long)(b * H + h) * W + wc, h < H ? W - wc : 0, 0, g, &bns, i);
      }
      __syncthreads();  // the strips are dead
      bns.template finish<4, 1>(p, reinterpret_cast<float*>(smem), wave, 0, lane, (int)blockIdx.x, 0, g);
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int h = h0 + 2 * wave + (i >> 1), wc = w0 + (i & 1) * 16;
        epi_pass_bf16<FN, FM, false>(p, acc[i], strip, lane, (long)(b * H + h) * W + wc, h < H ? W - wc : 0, 0, g);
      }
    }
    return;
  }
  const int hw = H * W;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int h = h0 + 2 * wave + (i >> 1);
    if (h >= H) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int w = w0 + (i & 1) * 16 + kg * 4 + r;
      if (w >= W) continue;
      const size_t m = (size_t)(b * H + h) * W + w;
#pragma unroll
      for (int n = 0; n < FN; ++n) {
        const int nn = n * 16 + p16;
        if (nn >= p.n_g) continue;
        const int ch = g * p.n_g + nn;
        float v = acc[i][n][r] * p.alpha;
        if (p.scale) v *= p.scale[ch];
        if (p.shift) v += p.shift[ch];
        if (res) v += bf2f(res[m * p.res_ld + ch]);
        v = apply_act(v, p.act);
        size_t o;
        if (p.dst_nchw) o = ((size_t)b * p.Cd + ch) * hw + (size_t)h * W + w;
        else o = m * p.dst_ld + ch;
        if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
        else reinterpret_cast<T*>(p.dst)[o] = f2bf(v);
      }
    }
  }
}

int halo_geo(const tfpp_conv_params& p) { return p.stride == 1 ? 0 : (p.mode == 0 ? 1 : 2); }

size_t halo_lds_bytes(const tfpp_conv_params& p, int fn) {
  const int ksteps = (9 * p.ks_g + 31) / 32;
  const int hpix = halo_geo(p) == 1 ? HaloGeo<1>::HH * HaloGeo<1>::HWID : HaloGeo<0>::HH * HaloGeo<0>::HWID;
  return (size_t)hpix * p.ks_g * 2 + (size_t)fn * 16 * (ksteps * 32 + 8) * 2;
}

template <int FN, int CV> int launch_halo(const tfpp_conv_params& p, hipStream_t st) {
  const int tiles_w = cdiv(p.Wd, TW), tiles_h = cdiv(p.Hd, TH), ksteps = (9 * p.ks_g + 31) / 32;
  const size_t lds = halo_lds_bytes(p, FN);
  dim3 grid((unsigned)(tiles_w * tiles_h * p.B), (unsigned)p.G);
  const int geo = halo_geo(p);
  if (geo != 0) {
    if constexpr (FN <= 2 && (CV == 0 || CV == 3)) {  // RegNet group width 24 (unrolled staging) or the run-time loop
      if (p.bns_partial) return TFPP_EINVAL;
      static unsigned long long attr_mask = 0;
      if (tfpp_first_use_on_this_device(&attr_mask)) {  // the 17 x 65 input halo of the stride-2 forward needs > 64 KB with 24 channels
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<FN, CV, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
      }
      if (p.in_bn.scale && geo != 1) return TFPP_EINVAL;
      if (geo == 1 && p.in_bn.scale) {
        static unsigned long long attr_mask_bn = 0;
        if (tfpp_first_use_on_this_device(&attr_mask_bn))
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<FN, CV, false, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, false, 1, true>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
      } else if (geo == 1) hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, false, 1>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
      else hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, false, 2>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
      TFPP_CHECK_LAUNCH();
      return 0;
    } else {
      return TFPP_EINVAL;  // conv_halo_supported does not route such shapes here
    }
  }
  if (p.in_bn.scale) {
    if (p.bns_partial || p.mode != 0 || p.ks_g > 64) return TFPP_EINVAL;
    hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, false, 0, true>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
  } else if (p.bns_partial) hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, true>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
  else hipLaunchKernelGGL((conv3x3_halo_kernel<FN, CV, false>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, ksteps);
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <int FN> int launch_halo_cv(const tfpp_conv_params& p, hipStream_t st) {
  if (p.stride != 1) return (p.ks_g >> 3) == 3 ? launch_halo<FN, 3>(p, st) : launch_halo<FN, 0>(p, st);
  switch (p.ks_g >> 3) {  // the channel counts of this model get the unrolled staging; anything else the run-time loop
    case 1: return launch_halo<FN, 1>(p, st);
    case 2: return launch_halo<FN, 2>(p, st);
    case 3: return launch_halo<FN, 3>(p, st);
    case 4: return launch_halo<FN, 4>(p, st);
    case 8: return launch_halo<FN, 8>(p, st);
    default: return launch_halo<FN, 0>(p, st);
  }
}
}  // namespace

// narrowest map the 8 x 32 tiles are used on (TFPP_HALO_MIN_W, default 16: the 16 x 16 LiDAR stage-3 maps run with half of every tile's
// columns masked -- still one staged pass instead of nine gathers; 32 = the rounds 1-5 behaviour)
static int halo_min_w() { static const int v = [] { const char* e = std::getenv("TFPP_HALO_MIN_W"); return e ? std::atoi(e) : 16; }(); return v; }

// variant code 300 + FN
bool conv_halo_supported(const tfpp_conv_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_CONV_HALO"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || dtype != TFPP_BF16) return false;
  if (p.R != 3 || p.S != 3 || p.pad != 1) return false;
  if (p.ks_g % 8 || p.src_ld % 8 || p.n_g > 64 || p.Wd < halo_min_w() || p.Hd < 4) return false;
  const int fn = p.n_g <= 16 ? 1 : (p.n_g <= 32 ? 2 : 4);
  if (p.stride == 2) {  // even feature maps only: forward Hs = 2 Hd, data gradient Hd = 2 Hs (TFPP_CONV_HALO_S2=0: implicit GEMM)
    static const int s2 = [] { const char* e = std::getenv("TFPP_CONV_HALO_S2"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool geo_ok = p.mode == 0 ? (p.Hs == 2 * p.Hd && p.Ws == 2 * p.Wd) : (p.Hd == 2 * p.Hs && p.Wd == 2 * p.Ws);
    return s2 && geo_ok && fn <= 2 && !p.bns_partial && halo_lds_bytes(p, fn) <= 120 * 1024;
  }
  if (p.stride != 1 || p.Hs != p.Hd || p.Ws != p.Wd) return false;
  static const int max_lds = [] { const char* e = std::getenv("TFPP_CONV_HALO_MAX_LDS"); return e ? std::atoi(e) : 65536; }();
  return halo_lds_bytes(p, fn) <= max_lds;
}

// normalise-on-load (tfpp_conv_params.in_bn): forward launches of this kernel with <= 64 input channels per group
bool conv_halo_in_bn_ok(const tfpp_conv_params& p, int dtype) {
  return conv_halo_supported(p, dtype) && p.mode == 0 && p.ks_g <= 64 && !p.bns_partial;
}

int conv_halo_variant(const tfpp_conv_params& p) { return 300 + (p.n_g <= 16 ? 1 : (p.n_g <= 32 ? 2 : 4)); }

int conv_halo_mtiles(const tfpp_conv_params& p) { return cdiv(p.Wd, TW) * cdiv(p.Hd, TH) * p.B; }

int conv_gemm_halo(const tfpp_conv_params& p, hipStream_t st) {
  switch (conv_halo_variant(p) - 300) {
    case 1: return launch_halo_cv<1>(p, st);
    case 2: return launch_halo_cv<2>(p, st);
    default: return launch_halo_cv<4>(p, st);
  }
}
