// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with small channel counts, bf16:
//     dW[n][tap * Cin_g + c] = sum_pixels dY[pix][n] * X[pix + tap][c]
// The implicit-GEMM weight-gradient kernels gather X once per tap, i.e. pull every input pixel through L2 nine times; on the
// full-resolution decoder layers (3.1 M pixels x 32 channels) and the RegNet grouped convs that traffic bounds them.  Here a
// workgroup walks 8 x 32 pixel tiles of one group: per tile it stages the (8+2) x (32+2) x Cin_g input halo and the 256 x N dY
// tile in LDS once -- both already "k-major" for this product (pixel = reduction index = slow index) -- and every tap is an
// LDS offset.  MFMA fragments (8 consecutive pixels of one row per lane) come from ds_read_b64_tr_b16.  The whole
// N x 9*Cin_g gradient of the group stays in registers across the tiles a workgroup visits (4 waves split the (tap, 16-channel)
// column blocks); at the end each workgroup stores one fp32 slice and wgrad_reduce_kernel sums the slices into dW.
#include "gemm_core.h"
#include "gemm_internal.h"
#include <cstdlib>

namespace {
constexpr int TH = 8, TW = 32, HH = TH + 2, HWID = TW + 2;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ uint2 lds_tr16(const bf16_t* p) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
  return make_uint2((unsigned)(unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16),
                    (unsigned)(unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16));
}

// FNN: 16-channel fragments of dY (n_g <= 16 FNN); CV = Cin_g / 8 (compile time: the staging loops are fully unrolled, all of a
// thread's global loads are in flight before its first LDS store); CB = 16-channel blocks of X per tap
// XBN (round 6): x exists only as (raw output of the 1x1 convolution in front, final BatchNorm scale / shift): every staged chunk of the halo
// becomes relu(raw * scale[c] + shift[c]) on its way into LDS; out-of-image chunks stay zero (tfpp_wgrad_params.x_scale).
template <int FNN, int CV, bool XBN = false>
__global__ __launch_bounds__(256) void wgrad3x3_halo_kernel(tfpp_wgrad_params p, int tiles_w, int tiles_h, int nblk, int xcd_order) {
  typedef bf16_t T;
  constexpr int CB = CV <= 2 ? 1 : (CV <= 4 ? 2 : 4);
  constexpr int NCOL = 9 * CB, NCW = (NCOL + 3) / 4;  // column blocks in total / per wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int cin = CV * 8, cv = CV;
  const int ng = p.n_g, nv = ng >> 3;
  const int npitch = FNN * 16 + 8;  // dY row pitch in elements (16-byte skew)
  // Workgroup id -> (slice, group).  The groups of a layer read 16 * CV-byte pieces of the same NHWC pixel rows (three groups share a 128-byte line
  // at 24 channels per group).  xcd_order (1-D grid, nblk a multiple of 8): XCD x owns the slices x, x + 8, ... and runs ALL groups of a slice back
  // to back, so the pixel rows of a tile enter ONE L2 once and serve every group (the 2-D grid dealt the groups of a tile over all eight XCDs:
  // 57 MB fetched for a stage-3 layer whose operands are 28 MB, profiles/r06_pmc_all_kernels_before_pin.txt).
  int g, slice;
  if (xcd_order) {
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    g = j % p.G;
    slice = (j / p.G) * 8 + xcd;
  } else {
    g = blockIdx.y;
    slice = blockIdx.x;
  }
  const int H = p.Hd, W = p.Wd;
  T* halo = reinterpret_cast<T*>(smem);                       // [HH*HWID][cin] (+ one chunk row of slack for the last block's overhang)
  T* dyl = halo + (HH * HWID + 4) * cin;                      // [TH*TW][npitch]
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x) + g * cin;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy) + g * ng;

  f32x4_t acc[FNN][NCW];
#pragma unroll
  for (int i = 0; i < FNN; ++i)
#pragma unroll
    for (int c = 0; c < NCW; ++c) acc[i][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int m16 = lane & 15, kg = lane >> 4;
  const int ntiles = tiles_w * tiles_h * p.B;
  __shared__ float x_sc[XBN ? 64 : 1], x_sh[XBN ? 64 : 1];
  if constexpr (XBN) {
    if (tid < cin) { x_sc[tid] = p.x_scale[g * cin + tid]; x_sh[tid] = p.x_shift[g * cin + tid]; }
    __syncthreads();
  }
  for (int t = slice; t < ntiles; t += nblk) {
    const int tw = t % tiles_w, t2 = t / tiles_w, th = t2 % tiles_h, b = t2 / tiles_h;
    const int h0 = th * TH, w0 = tw * TW;
    // ---- stage X halo and dY tile (zero outside the image)
    constexpr int HCH = HH * HWID * CV, HIT = (HCH + 255) / 256, DCH = TH * TW * FNN * 2, DIT = DCH / 256;
    uint4 hv[HIT], dv[DIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int q = tid + it * 256, pix = q / cv, c = q - pix * cv;
      const int hr = pix / HWID, hc = pix - hr * HWID;
      const int h = h0 + hr - 1, w = w0 + hc - 1;
      hv[it] = make_uint4(0, 0, 0, 0);
      if (q < HCH && h >= 0 && h < H && w >= 0 && w < W) hv[it] = *reinterpret_cast<const uint4*>(x + ((size_t)(b * H + h) * W + w) * p.x_ld + c * 8);
    }
    unsigned inside_bits = 0u;
    if constexpr (XBN) {
      static_assert(HIT <= 32, "one validity bit per staged chunk");
#pragma unroll
      for (int it = 0; it < HIT; ++it) {
        const int q = tid + it * 256, pix = q / cv;
        const int hr = pix / HWID, hc = pix - hr * HWID;
        const int h = h0 + hr - 1, w = w0 + hc - 1;
        if (q < HCH && h >= 0 && h < H && w >= 0 && w < W) inside_bits |= 1u << it;
      }
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int q = tid + it * 256, pix = q / (FNN * 2), c = q - pix * (FNN * 2);
      const int h = h0 + pix / TW, w = w0 + pix % TW;
      dv[it] = make_uint4(0, 0, 0, 0);
      if (c < nv && h < H && w < W) dv[it] = *reinterpret_cast<const uint4*>(dy + ((size_t)(b * H + h) * W + w) * p.dy_ld + c * 8);
    }
    if constexpr (XBN) {
#pragma unroll
      for (int it = 0; it < HIT; ++it) {
        if (!(inside_bits & (1u << it))) continue;
        const int c8 = ((tid + it * 256) % cv) * 8;
        float f[8];
        unpack16<T>(hv[it], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float tv = f[e] * x_sc[c8 + e] + x_sh[c8 + e];
          f[e] = p.x_relu ? (tv > 0.f ? tv : 0.f) : tv;
        }
        hv[it] = pack16<T>(f);
      }
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) { const int q = tid + it * 256; if (q < HCH) *reinterpret_cast<uint4*>(halo + (size_t)q * 8) = hv[it]; }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int q = tid + it * 256, pix = q / (FNN * 2), c = q - pix * (FNN * 2);
      *reinterpret_cast<uint4*>(dyl + (size_t)pix * npitch + c * 8) = dv[it];
    }
    __syncthreads();
    // ---- 8 K-steps: step ks = tile row ks (32 pixels); lane (m16, kg) feeds pixel columns kg*8 + (m16>>2) [+4], channel quad m16&3
#pragma unroll 2
    for (int ks = 0; ks < TH; ++ks) {
      Frag<T> fa[FNN];
      const int prow = ks * TW + kg * 8 + (m16 >> 2);
#pragma unroll
      for (int i = 0; i < FNN; ++i) {
        const T* a0 = dyl + (size_t)prow * npitch + i * 16 + (m16 & 3) * 4;
        const uint2 lo = lds_tr16(a0), hi = lds_tr16(a0 + 4 * npitch);
        fa[i].v = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int c = 0; c < NCW; ++c) {
        const int col = wave + 4 * c;  // column block (tap, cb) of this wave
        if (col < NCOL) {
          const int tap = col / CB, cb = col - tap * CB, r = tap / 3, s = tap - r * 3;
          const T* b0 = halo + (size_t)((ks + r) * HWID + kg * 8 + (m16 >> 2) + s) * cin + cb * 16 + (m16 & 3) * 4;
          const uint2 lo = lds_tr16(b0), hi = lds_tr16(b0 + 4 * cin);
          Frag<T> fb;
          fb.v = make_uint4(lo.x, lo.y, hi.x, hi.y);
#pragma unroll
          for (int i = 0; i < FNN; ++i) frag_mma(fa[i], fb, acc[i][c]);
        }
      }
    }
    __syncthreads();  // before the next tile overwrites the stages
  }

  // ---- this workgroup's slice -> workspace [blockIdx.x][G*n_g][KK]
  const int KK = 9 * cin;
  float* __restrict__ wsp = p.ws + ((size_t)slice * p.G * ng + (size_t)g * ng) * KK;
#pragma unroll
  for (int c = 0; c < NCW; ++c) {
    const int col = wave + 4 * c;
    if (col >= NCOL) continue;
    const int tap = col / CB, cb = col - tap * CB, ch = cb * 16 + m16;
    if (ch >= cin) continue;
    const int kk = tap * cin + ch;
#pragma unroll
    for (int i = 0; i < FNN; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = i * 16 + kg * 4 + r;
        if (n < ng) wsp[(size_t)n * KK + kk] = acc[i][c][r];
      }
  }
}

size_t halo_lds_bytes(const tfpp_wgrad_params& p, int fnn) {
  return (size_t)(HH * HWID + 4) * p.ks_g * 2 + (size_t)TH * TW * (fnn * 16 + 8) * 2;
}

template <int FNN, int CV> int launch(const tfpp_wgrad_params& p, int nblk, hipStream_t st) {
  const int tiles_w = cdiv(p.Wd, TW), tiles_h = cdiv(p.Hd, TH);
  const size_t lds = halo_lds_bytes(p, FNN);
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_halo_kernel<FNN, CV>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_halo_kernel<FNN, CV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  }
  static const int xo_env = [] { const char* e = std::getenv("TFPP_WGRAD_HALO_XCD"); return (e && e[0] == '0') ? 0 : 1; }();
  const int xo = (xo_env && p.G > 1 && nblk % 8 == 0) ? 1 : 0;
  const dim3 grid = xo ? dim3((unsigned)(nblk * p.G)) : dim3((unsigned)nblk, (unsigned)p.G);
  if (p.x_scale) hipLaunchKernelGGL((wgrad3x3_halo_kernel<FNN, CV, true>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, nblk, xo);
  else hipLaunchKernelGGL((wgrad3x3_halo_kernel<FNN, CV>), grid, dim3(256), lds, st, p, tiles_w, tiles_h, nblk, xo);
  TFPP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

// narrowest map the 8 x 32 tiles are used on (TFPP_HALO_MIN_W, default 16: the 16 x 16 LiDAR stage-3 maps run with half of every tile's
// columns masked -- still one staged pass instead of nine gathers; 32 = the rounds 1-5 behaviour)
static int halo_min_w() { static const int v = [] { const char* e = std::getenv("TFPP_HALO_MIN_W"); return e ? std::atoi(e) : 16; }(); return v; }

// number of workgroups (= workspace slices) per group; 0 if the kernel does not apply
int wgrad_halo_slices(const tfpp_wgrad_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_WGRAD_HALO"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || dtype != TFPP_BF16 || !p.ws) return 0;
  if (p.R != 3 || p.S != 3 || p.stride != 1 || p.pad != 1 || p.Hs != p.Hd || p.Ws != p.Wd) return 0;
  if (p.ks_g % 8 || p.n_g % 8 || p.x_ld % 8 || p.dy_ld % 8 || p.n_g > 64 || p.ks_g > 64 || p.Wd < halo_min_w() || p.Hd < 4) return 0;
  if (((uintptr_t)p.dy & 15) || ((uintptr_t)p.x & 15)) return 0;
  const int cv8 = p.ks_g >> 3;
  if (!(cv8 == 1 || cv8 == 2 || cv8 == 3 || cv8 == 4 || cv8 == 8)) return 0;  // instantiated channel counts
  const int fnn = p.n_g <= 16 ? 1 : (p.n_g <= 32 ? 2 : 4);
  if (halo_lds_bytes(p, fnn) > 98304) return 0;
  const long tiles = (long)cdiv(p.Wd, TW) * cdiv(p.Hd, TH) * p.B;
  if (p.ks_g > 32 && tiles < 2048) return 0;  // 64-channel inputs on small maps: the LDS-DMA implicit GEMM measured faster
  // workgroups over all groups (each writes one fp32 slice of its group's gradient: fewer workgroups = fewer slice bytes written and
  // summed, more tiles walked one after the other per workgroup).  TFPP_WGRAD_HALO_WGS, default 512 (1024 / 512 / 288 / 2048: 23.36 / 23.26 / 23.29 / 23.35 ms per step)
  static const long total_wgs = [] { const char* e = std::getenv("TFPP_WGRAD_HALO_WGS"); const long v = e ? std::atol(e) : 512; return v < 1 ? 1 : v; }();
  long nblk = total_wgs / p.G;
  if (nblk < 1) nblk = 1;
  if (nblk > tiles) nblk = tiles;
  if (p.G > 1 && nblk >= 8) nblk = (nblk + 4) / 8 * 8 <= tiles ? (nblk + 4) / 8 * 8 : nblk / 8 * 8;  // whole XCD rounds of slices (see the kernel)
  const long slice = (long)p.G * p.n_g * 9 * p.ks_g;
  if (nblk * slice > p.ws_floats) nblk = p.ws_floats / slice;
  return nblk >= 1 ? (int)nblk : 0;
}

int conv_wgrad_halo(const tfpp_wgrad_params& p, int nblk, hipStream_t st) {
  const int fnn = p.n_g <= 16 ? 1 : (p.n_g <= 32 ? 2 : 4), cv = p.ks_g >> 3;
#define HALO_CASE(F, C) if (fnn == F && cv == C) return launch<F, C>(p, nblk, st)
  HALO_CASE(1, 1); HALO_CASE(1, 2); HALO_CASE(1, 3); HALO_CASE(1, 4); HALO_CASE(1, 8);
  HALO_CASE(2, 1); HALO_CASE(2, 2); HALO_CASE(2, 3); HALO_CASE(2, 4); HALO_CASE(2, 8);
  HALO_CASE(4, 1); HALO_CASE(4, 2); HALO_CASE(4, 3); HALO_CASE(4, 4); HALO_CASE(4, 8);
#undef HALO_CASE
  return TFPP_EINVAL;
}
