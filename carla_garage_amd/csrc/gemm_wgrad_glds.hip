// Weight gradient with a multi-stage LDS-DMA ring, bf16:  dW[n][kk] += sum_pixels dY[pix][n] * Xgather[pix][kk].
//
// Both operands are "k-major" for this product (the reduction index, the pixel, is the slow index of the NHWC tensors), so a
// 32-pixel stage of dY (BM channels) and of X (BN gathered (tap, channel) columns) is copied HBM/L2 -> LDS by
// global_load_lds_dwordx4 exactly as it lies in memory -- [pixel][channel chunk] -- and the MFMA fragments (8 consecutive
// pixels per lane) are produced by the hardware transpose read ds_read_b64_tr_b16.  NSTAGE stages in flight, one raw
// s_barrier per stage, counted vmcnt (same skeleton as gemm_glds.hip).  LDS-DMA writes lane-linear, so the image is the
// linear [pixel][CH x 16 B] and the bank-conflict swizzle is applied to the SOURCE chunk index and again on the read:
//     slot of (pixel p, chunk c) = c ^ F(p),   F(p) = 2*((p>>1)&1) + 4*((p>>3)&1)          (64-channel rows, 128 B)
// A transpose read takes, per 16-lane group, 4 pixel rows x 32 B; within the 32 lanes the hardware serves together the rows
// are {0..3, 8..11} (+16): rows of equal parity alias in the 64 banks and F moves them to distinct 32-byte pairs.
// The single-buffered kernel in gemm_kernels.hip (load -> ds_write -> barrier -> MFMA -> barrier per 64 pixels) measured
// 25-35 us of main loop on the 576x576 / 216x216 layers; this one keeps NSTAGE-1 stages of loads behind the MFMAs.
#include "gemm_core.cuh"
#include "gemm_internal.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
static __device__ __attribute__((aligned(16))) unsigned int tfpp_zero_page[4] = {0u, 0u, 0u, 0u};  // LDS-DMA cannot zero-fill

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_t lds_read_tr16_b64_asm(unsigned addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

namespace {
constexpr int BM = 64, BN = 64, BKP = 32;          // tile: 64 output rows (n) x 64 columns (kk), 32 pixels per stage
constexpr int ROW_BYTES = 128, CH = 8;             // 64 channels x 2 B per pixel row, 16-byte chunks per row
constexpr int A_BYTES = BKP * ROW_BYTES, STAGE_BYTES = 2 * A_BYTES;

__device__ __forceinline__ int swz(int p) { return 2 * ((p >> 1) & 1) + 4 * ((p >> 3) & 1); }

template <int NSTAGE>
__global__ __launch_bounds__(256) void conv_wgrad_glds_kernel(tfpp_wgrad_params p) {
  typedef bf16_t T;
  constexpr int FM = 2, FN = 2, LOADS = 2;  // 2x2 waves, wave tile 32 x 32; one A and one B DMA per wave per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int KK = p.R * p.S * p.ks_g;
  // 1-D grid.  Workgroup id -> XCD id % 8.  With the slices a multiple of 8, XCD x walks slices x, x+8, ... and, inside a slice, all
  // (n, kk) tiles back to back: the slice's dY / X pixel slab (1-2 MB) is fetched into that XCD's L2 once and shared by every tile.
  // (x = slice fastest, as the LDS-staged kernel orders them, spreads the tiles of a slab over the whole launch: measured 2.6x
  // the algorithmic bytes at the fabric.)
  const int tiles_m = (p.n_g + BM - 1) / BM, tiles_n = (KK + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  int g, split, tile;
  {
    const int id = blockIdx.x, per_g = p.splits * ntiles;
    g = id / per_g;
    const int r = id - g * per_g;
    if ((p.splits & 7) == 0) {
      const int xcd = r & 7, j = r >> 3;
      tile = j % ntiles;
      split = (j / ntiles) * 8 + xcd;
    } else {
      split = r % p.splits;
      tile = r / p.splits;
    }
  }
  const int bm0 = (tile % tiles_m) * BM, bn0 = (tile / tiles_m) * BN;
  const long P = (long)p.B * p.Hd * p.Wd;
  const long per = ((P + p.splits - 1) / p.splits + BKP - 1) / BKP * BKP;
  const long p_beg = (long)split * per, p_end = (p_beg + per < P) ? p_beg + per : P;
  const int nst = p_end > p_beg ? (int)((p_end - p_beg + BKP - 1) / BKP) : 0;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy) + g * p.n_g;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x) + g * p.ks_g;
  const T* zero = reinterpret_cast<const T*>(tfpp_zero_page);
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;

  // this thread's chunk of every stage: slot q = tid -> pixel row pk = q / 8, physical chunk cp = q % 8, logical chunk cp ^ F(pk)
  const int pk = tid >> 3, cl = (tid & 7) ^ swz(pk);
  const int a_n = bm0 + cl * 8;   // dY channel of the chunk
  const int b_kk = bn0 + cl * 8;  // gathered column (tap, channel) of the chunk
  const bool a_ok = a_n < p.n_g, b_ok = b_kk < KK;
  const int b_rs = b_ok ? b_kk / p.ks_g : 0, b_c = b_kk - b_rs * p.ks_g, b_r = b_rs / p.S, b_s = b_rs - b_r * p.S;
  const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);
  const int hw = p.Hd * p.Wd;

  auto issue = [&](int st) {  // LDS-DMA of pixel stage st into ring slot st % NSTAGE
    const unsigned stage = lds_base + (unsigned)((st % NSTAGE) * STAGE_BYTES);
    const long pix = p_beg + (long)st * BKP + pk;
    const T* ga = zero;
    const T* gb = zero;
    if (pix < p_end) {
      if (a_ok) ga = dy + (size_t)pix * p.dy_ld + a_n;
      if (b_ok) {
        if (pointwise) gb = x + (size_t)pix * p.x_ld + b_c;
        else {
          const int b = (int)(pix / hw), rem = (int)(pix - (long)b * hw), hd = rem / p.Wd, wd = rem - hd * p.Wd;
          const int hs = hd * p.stride - p.pad + b_r, ws = wd * p.stride - p.pad + b_s;
          if (hs >= 0 && hs < p.Hs && ws >= 0 && ws < p.Ws) gb = x + ((size_t)(b * p.Hs + hs) * p.Ws + ws) * p.x_ld + b_c;
        }
      }
    }
    __builtin_amdgcn_global_load_lds((gbl_void_t*)ga, (lds_void_t*)(stage + (unsigned)(wave * 1024)), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void_t*)gb, (lds_void_t*)(stage + (unsigned)(A_BYTES + wave * 1024)), 16, 0, 0);
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transpose-read offsets (fixed): lane (m = l & 15, kg = l >> 4) addresses pixel row kg*8 + (m >> 2) [+4 for the second read],
  // 8 bytes at channel quad (frag_row0 / 4 + (m & 3)) -> chunk (frag_row0 / 8 + (m & 3) / 2), half (m & 3) & 1
  const int m16 = lane & 15, kg = lane >> 4;
  const int prow = kg * 8 + (m16 >> 2);
  unsigned a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int c = (wm * 32 + i * 16) / 8 + ((m16 & 3) >> 1);
    a_off[i] = (unsigned)(prow * ROW_BYTES + ((c ^ swz(prow)) * 16) + (m16 & 1) * 8);
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = (wn * 32 + j * 16) / 8 + ((m16 & 3) >> 1);
    b_off[j] = (unsigned)(A_BYTES + prow * ROW_BYTES + ((c ^ swz(prow)) * 16) + (m16 & 1) * 8);
  }

#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nst) issue(s);

  for (int st = 0; st < nst; ++st) {
    const int ahead = (nst - 1 - st) < (NSTAGE - 2) ? (nst - 1 - st) : (NSTAGE - 2);
    switch (ahead) {  // wave-uniform
#define TFPP_WAIT_CASE(A) case A: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * LOADS) : "memory"); break
      TFPP_WAIT_CASE(0); TFPP_WAIT_CASE(1); TFPP_WAIT_CASE(2); TFPP_WAIT_CASE(3); TFPP_WAIT_CASE(4); TFPP_WAIT_CASE(5); TFPP_WAIT_CASE(6);
#undef TFPP_WAIT_CASE
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // stage st landed for every wave; every wave is done with stage st-1
    if (st + NSTAGE - 1 < nst) issue(st + NSTAGE - 1);
    const unsigned stage = lds_base + (unsigned)((st % NSTAGE) * STAGE_BYTES);
    Frag<T> fa[FM], fb[FN];
    u32x2_t lo[FM + FN], hi[FM + FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) { lo[i] = lds_read_tr16_b64_asm(stage + a_off[i]); hi[i] = lds_read_tr16_b64_asm(stage + a_off[i] + 4 * ROW_BYTES); }
#pragma unroll
    for (int j = 0; j < FN; ++j) { lo[FM + j] = lds_read_tr16_b64_asm(stage + b_off[j]); hi[FM + j] = lds_read_tr16_b64_asm(stage + b_off[j] + 4 * ROW_BYTES); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FM; ++i) fa[i].v = make_uint4(lo[i][0], lo[i][1], hi[i][0], hi[i][1]);
#pragma unroll
    for (int j = 0; j < FN; ++j) fb[j].v = make_uint4(lo[FM + j][0], lo[FM + j][1], hi[FM + j][0], hi[FM + j][1]);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) frag_mma(fa[i], fb[j], acc[i][j]);
  }

  // ---- epilogue: slice -> workspace [split][G*n_g][KK] (summed by wgrad_reduce_kernel), or straight into dw
  const int RS = p.R * p.S;
  if (p.ws && p.splits > 1) {
    float* __restrict__ wsp = p.ws + ((size_t)split * p.G * p.n_g + (size_t)g * p.n_g) * KK;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = bm0 + wm * 32 + i * 16 + kg * 4 + r;
        if (n >= p.n_g) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int kk = bn0 + wn * 32 + j * 16 + m16;
          if (kk < KK) wsp[(size_t)n * KK + kk] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = bm0 + wm * 32 + i * 16 + kg * 4 + r;
      if (n >= p.n_g) continue;
      int row = g * p.n_g + n;
      if (p.row_map) row = p.row_map[row];
      if (row < 0) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int kk = bn0 + wn * 32 + j * 16 + m16;
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
        else *o += acc[i][j][r];
      }
    }
}
}  // namespace

bool wgrad_glds_supported(const tfpp_wgrad_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_WGRAD_GLDS"); return (e && e[0] == '0') ? 0 : 1; }();
  const int KK = p.R * p.S * p.ks_g;
  return on && dtype == TFPP_BF16 && p.n_g > 32 && KK > 32 && p.n_g % 8 == 0 && p.ks_g % 8 == 0 && p.dy_ld % 8 == 0 && p.x_ld % 8 == 0 &&
         ((uintptr_t)p.dy & 15) == 0 && ((uintptr_t)p.x & 15) == 0;
}

int conv_wgrad_glds(const tfpp_wgrad_params& p, hipStream_t st) {
  constexpr int NSTAGE = 4;
  const int KK = p.R * p.S * p.ks_g;
  dim3 grid((unsigned)((long)p.G * p.splits * cdiv(p.n_g, BM) * cdiv(KK, BN)));
  hipLaunchKernelGGL(conv_wgrad_glds_kernel<NSTAGE>, grid, dim3(256), (size_t)NSTAGE * STAGE_BYTES, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}
