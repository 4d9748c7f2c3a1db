// Weight gradient with a multi-stage LDS-DMA ring, bf16:  dW[n][kk] += sum_pixels dY[pix][n] * Xgather[pix][kk].
//
// Both operands are "k-major" for this product (the reduction index, the pixel, is the slow index of the NHWC tensors), so a
// BKP-pixel stage of dY (TM channels) and of X (TN gathered (tap, channel) columns) is copied HBM/L2 -> LDS by
// global_load_lds_dwordx4 exactly as it lies in memory -- [pixel][channel chunk] -- and the MFMA fragments (8 consecutive
// pixels per lane) are produced by the hardware transpose read ds_read_b64_tr_b16.  NSTAGE stages in flight, one raw
// s_barrier per stage, counted vmcnt (same skeleton as gemm_glds.hip).  LDS-DMA writes lane-linear, so the image is the
// linear [pixel][CH x 16 B] and the bank-conflict swizzle is applied to the SOURCE chunk index and again on the read:
//     64-channel rows (128 B):  slot of (pixel p, chunk c) = c ^ (2*((p>>1)&1) + 4*((p>>3)&1))
//     128-channel rows (256 B): slot of (pixel p, chunk c) = c ^ (2*(p&3)      + 8*((p>>3)&1))
// A transpose read takes, per 16-lane group, 4 pixel rows x 32 B; the 32 lanes the hardware serves together address the rows
// {0..3, 8..11} (+16) at one 32-byte column: un-swizzled they alias in the 64 banks (8-way on 256-byte rows, which are exactly one
// bank row each); the XOR moves the eight rows to eight distinct 32-byte pairs.  Adding 4 or 32 to p (second read of a fragment,
// second k-step of a stage) does not change the swizzle, so one base offset per fragment serves all of its reads.
//
// Two instantiations:
//   64 x 64 tile, 4 waves (2x2), 32 pixels per stage, 4 stages: the small / medium layers, pixel reduction split over many
//     workgroups (slices summed by wgrad_reduce_kernel);
//   128 x 128 tile, 8 waves (2x4), 64 pixels per stage (two MFMA k-steps per barrier), 3 stages = 96 KB: the wide layers
//     (fusion linears 6048 x 1512: 564 tiles, no pixel split, the tile is added straight into dW; RegNet stage 3/4 1x1 convs with a
//     moderate split).  Twice the operand reuse per byte brought into LDS and 16 instead of 4 MFMAs per wave between barriers.
// The single-buffered kernel in gemm_kernels.hip (load -> ds_write -> barrier -> MFMA -> barrier per 64 pixels) measured
// 25-35 us of main loop on the 576x576 / 216x216 layers; this one keeps NSTAGE-1 stages of loads behind the MFMAs.
#include "gemm_core.h"
#include "gemm_internal.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
static __device__ __attribute__((aligned(16))) unsigned int tfpp_zero_page[4] = {0u, 0u, 0u, 0u};  // LDS-DMA cannot zero-fill

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ u32x2_t lds_read_tr16_b64_asm(unsigned addr) {  // OFF: 16-bit immediate byte offset
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

namespace {
template <int T> __device__ __forceinline__ int swz(int p);
template <> __device__ __forceinline__ int swz<64>(int p) { return 2 * ((p >> 1) & 1) + 4 * ((p >> 3) & 1); }
template <> __device__ __forceinline__ int swz<128>(int p) { return 2 * (p & 3) + 8 * ((p >> 3) & 1); }
template <> __device__ __forceinline__ int swz<256>(int p) { return 2 * (p & 3) + 8 * ((p >> 3) & 1); }  // 512-byte rows = two bank rows: the same eight 32-byte pairs

// PW: pointwise layers (every linear / 1x1 conv: all but the 3x3 heads) take a lean pixel loop -- scalar loop control, 32-bit running byte
// offsets against the uniform operand bases, out-of-tile channels clamped into the row instead of redirected to the zero page (they only
// feed gradient rows / columns that are never stored), the zero page only for the pixel tail of the last stage.  The general loop
// spends ~110 non-MFMA instructions per 16 MFMAs (run-time pointwise / tap branches, 64-bit pointer selects).
// One workgroup's share of one weight gradient: workgroup `wg_id` of the tfpp_conv_wgrad launch geometry of p (G * splits * tiles
// workgroups).  Called once per workgroup by conv_wgrad_glds_kernel and in a loop by the grouped kernel below.
// pin >= 0 (grouped launches, G = 1, layers with a small tile grid): the layer's p.splits pixel slices are UNITS pinned to XCDs -- unit u
// runs all of its tiles on XCD (pin + u) % 8, back to back, so the (tiles_m + tiles_n) operand panels of a slice enter ONE L2 once and are
// shared by all tiles of the slice (the host picks the slices so that this slab fits the L2).  The workgroup range of such a layer is
// 8 * ceil(splits / 8) * ntiles wide; workgroups whose (XCD, position) maps to no unit return at once.
template <int TM, int TN, int WGM, int WGN, int BKP, int NSTAGE, bool PW>
__device__ __forceinline__ void wgrad_glds_tile(const tfpp_wgrad_params& p, const int wg_id, const int pin = -1) {
  typedef bf16_t T;
  constexpr int NT = WGM * WGN * 64, NWAVES = WGM * WGN;
  constexpr int WM = TM / WGM, WN = TN / WGN, FM = WM / 16, FN = WN / 16, KS = BKP / 32;
  constexpr int ROW_A = TM * 2, ROW_B = TN * 2, CH_A = TM / 8, CH_B = TN / 8;  // bytes / 16-byte chunks per pixel row
  constexpr int A_BYTES = BKP * ROW_A, B_BYTES = BKP * ROW_B, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = A_BYTES / 16 / NT, B_INST = B_BYTES / 16 / NT, LOADS = A_INST + B_INST;
  static_assert(A_INST * 16 * NT == A_BYTES && B_INST * 16 * NT == B_BYTES, "stage must divide over the threads");
  static_assert(LOADS * (NSTAGE - 2) < 64, "vmcnt is a 6-bit counter");
  static_assert(KS <= 2 && (KS - 1) * (FM + FN) * 2 <= 15, "lgkmcnt ladder: one or two k-steps per stage, 4-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int KK = p.R * p.S * p.ks_g;
  // 1-D grid.  Workgroup id -> XCD id % 8.
  //  * pixel slices a multiple of 8: XCD x walks slices x, x+8, ... and, inside a slice, all (n, kk) tiles back to back: the slice's
  //    dY / X pixel slab (1-2 MB) is fetched into that XCD's L2 once and shared by every tile (x = slice fastest, as the LDS-staged
  //    kernel orders them, spreads the tiles of a slab over the whole launch: measured 2.6x the algorithmic bytes at the fabric);
  //  * otherwise (one slice, or a few): XCD x owns every 8th tile along the LONGER tile axis and walks the shorter axis back to
  //    back, so an XCD fetches 1/8 of the large operand plus the small one (each XCD computing a compact block of the tile grid is
  //    the best 8 separate L2s allow: ~2.3x the algorithmic bytes at 128x128 tiles, absorbed by the Infinity Cache).
  const int tiles_m = (p.n_g + TM - 1) / TM, tiles_n = (KK + TN - 1) / TN, ntiles = tiles_m * tiles_n;
  int g, split, tile_m, tile_n;
  {
    const int id = wg_id, per_g = p.splits * ntiles;
    g = pin >= 0 ? 0 : id / per_g;  // (pinned layers have one group and a range wider than splits * ntiles)
    const int r = id - g * per_g;
    int tile;
    if (pin >= 0) {
      const int xcd = r & 7, j = r >> 3, rnd = j / ntiles;
      split = rnd * 8 + ((xcd - pin) & 7);
      if (split >= p.splits) return;  // (uniform over the workgroup)
      tile = j - rnd * ntiles;
      tile_m = tile % tiles_m;
      tile_n = tile / tiles_m;
    } else if ((p.splits & 7) == 0) {
      const int xcd = r & 7, j = r >> 3;
      tile = j % ntiles;
      split = (j / ntiles) * 8 + xcd;
      tile_m = tile % tiles_m;
      tile_n = tile / tiles_m;
    } else {
      split = r % p.splits;
      tile = r / p.splits;
      const bool m_long = tiles_m >= tiles_n;
      const int nl = m_long ? tiles_m : tiles_n, ns = m_long ? tiles_n : tiles_m;  // long / short axis
      const int full = (nl >> 3) << 3;
      int tl, ts;
      if (tile < full * ns) {
        const int xcd = tile & 7, j = tile >> 3;
        ts = j % ns;
        tl = (j / ns) * 8 + xcd;
      } else {  // the last (< 8) tiles of the long axis keep the plain order
        const int q = tile - full * ns;
        ts = q % ns;
        tl = full + q / ns;
      }
      tile_m = m_long ? tl : ts;
      tile_n = m_long ? ts : tl;
    }
  }
  if constexpr (PW) {  // the divisions above run on the vector ALU: the results are wave-uniform (scalar loop control below)
    g = __builtin_amdgcn_readfirstlane(g); split = __builtin_amdgcn_readfirstlane(split);
    tile_m = __builtin_amdgcn_readfirstlane(tile_m); tile_n = __builtin_amdgcn_readfirstlane(tile_n);
  }
  const int bm0 = tile_m * TM, bn0 = tile_n * TN;
  const long P = (long)p.B * p.Hd * p.Wd;
  const long per = ((P + p.splits - 1) / p.splits + BKP - 1) / BKP * BKP;
  const long p_beg = (long)split * per, p_end = (p_beg + per < P) ? p_beg + per : P;
  const int nst = p_end > p_beg ? (int)((p_end - p_beg + BKP - 1) / BKP) : 0;
  const T* __restrict__ dy = reinterpret_cast<const T*>(p.dy) + g * p.n_g;
  const T* __restrict__ x = reinterpret_cast<const T*>(p.x) + g * p.ks_g;
  const T* zero = reinterpret_cast<const T*>(tfpp_zero_page);
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;

  const int hw = p.Hd * p.Wd;
  const int npix = (int)(p_end - p_beg);             // pixels of this slice (> 0 when nst > 0)
  const int tail = npix - (nst - 1) * BKP;           // valid pixel rows of the last stage (1 .. BKP)
  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transpose-read offsets (fixed): lane (m = l & 15, kg = l >> 4) addresses pixel row kg*8 + (m >> 2) [+4 for the second read,
  // +32 per k-step], 8 bytes at channel quad (frag_row0 / 4 + (m & 3)) -> chunk (frag_row0 / 8 + (m & 3) / 2), half (m & 3) & 1
  const int m16 = lane & 15, kg = lane >> 4;
  const int prow = kg * 8 + (m16 >> 2);
  unsigned a_off[FM], b_off[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int c = (wm * WM + i * 16) / 8 + ((m16 & 3) >> 1);
    a_off[i] = (unsigned)(prow * ROW_A + ((c ^ swz<TM>(prow)) * 16) + (m16 & 1) * 8);
  }
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int c = (wn * WN + j * 16) / 8 + ((m16 & 3) >> 1);
    b_off[j] = (unsigned)(A_BYTES + prow * ROW_B + ((c ^ swz<TN>(prow)) * 16) + (m16 & 1) * 8);
  }

  auto stage_mma = [&](unsigned stage) {
    // all transpose reads of the stage are issued up front (DS operations retire in order): the MFMAs of k-step ks start once its
    // own (FM + FN) * 2 reads have landed, while the reads of the later k-steps are still in flight
    u32x2_t lo[KS][FM + FN], hi[KS][FM + FN];
    unsigned a_adr[FM], b_adr[FN];  // one VALU add per fragment; the k-step / second-read displacements are instruction immediates
#pragma unroll
    for (int i = 0; i < FM; ++i) a_adr[i] = stage + a_off[i];
#pragma unroll
    for (int j = 0; j < FN; ++j) b_adr[j] = stage + b_off[j];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      lo[0][i] = lds_read_tr16_b64_asm<0>(a_adr[i]);
      hi[0][i] = lds_read_tr16_b64_asm<4 * ROW_A>(a_adr[i]);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      lo[0][FM + j] = lds_read_tr16_b64_asm<0>(b_adr[j]);
      hi[0][FM + j] = lds_read_tr16_b64_asm<4 * ROW_B>(b_adr[j]);
    }
    if constexpr (KS == 2) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        lo[1][i] = lds_read_tr16_b64_asm<32 * ROW_A>(a_adr[i]);
        hi[1][i] = lds_read_tr16_b64_asm<36 * ROW_A>(a_adr[i]);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        lo[1][FM + j] = lds_read_tr16_b64_asm<32 * ROW_B>(b_adr[j]);
        hi[1][FM + j] = lds_read_tr16_b64_asm<36 * ROW_B>(b_adr[j]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      constexpr int PER = (FM + FN) * 2;
      if (ks == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PER * (KS - 1)) : "memory");  // the reads of the later k-step may stay in flight
      __builtin_amdgcn_sched_barrier(0);
      Frag<T> fa[FM], fb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) fa[i].v = make_uint4(lo[ks][i][0], lo[ks][i][1], hi[ks][i][0], hi[ks][i][1]);
#pragma unroll
      for (int j = 0; j < FN; ++j) fb[j].v = make_uint4(lo[ks][FM + j][0], lo[ks][FM + j][1], hi[ks][FM + j][0], hi[ks][FM + j][1]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) frag_mma(fa[i], fb[j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (PW) {
    const int nst_u = __builtin_amdgcn_readfirstlane(nst);
    const char* a_base = reinterpret_cast<const char*>(dy);
    const char* b_base = reinterpret_cast<const char*>(x);
    unsigned a_vo[A_INST], b_vo[B_INST];
    bool a_tz[A_INST], b_tz[B_INST];  // pixel row beyond the slice in the last stage
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int q = i * NT + tid, pk = q / CH_A;
      int n = bm0 + (((q % CH_A) ^ swz<TM>(pk)) * 8);
      n = n < p.n_g ? n : p.n_g - 8;  // (n_g % 8 == 0) finite data of this row; feeds gradient rows that are never stored
      a_vo[i] = ((unsigned)(p_beg + pk) * (unsigned)p.dy_ld + (unsigned)n) * 2u;
      a_tz[i] = pk >= tail;
    }
#pragma unroll
    for (int j = 0; j < B_INST; ++j) {
      const int q = j * NT + tid, pk = q / CH_B;
      int kk = bn0 + (((q % CH_B) ^ swz<TN>(pk)) * 8);
      kk = kk < KK ? kk : KK - 8;
      b_vo[j] = ((unsigned)(p_beg + pk) * (unsigned)p.x_ld + (unsigned)kk) * 2u;
      b_tz[j] = pk >= tail;
    }
    const unsigned a_step = (unsigned)BKP * (unsigned)p.dy_ld * 2u, b_step = (unsigned)BKP * (unsigned)p.x_ld * 2u;
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    const unsigned ring_end = lds_base + NSTAGE * STAGE_BYTES;
    auto issue = [&](unsigned stage, bool last) {  // `last` is wave-uniform; both variants issue LOADS loads per thread, in the same order
      if (!last) {
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_base + a_vo[i]), (lds_void_t*)(stage + (i * NWAVES + wave_u) * 1024u), 16, 0, 0);
          a_vo[i] += a_step;
        }
#pragma unroll
        for (int j = 0; j < B_INST; ++j) {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_base + b_vo[j]), (lds_void_t*)(stage + A_BYTES + (j * NWAVES + wave_u) * 1024u), 16, 0, 0);
          b_vo[j] += b_step;
        }
      } else {  // pixel rows past the end of the slice contribute nothing: both operands read the zero page there
        const char* zp = reinterpret_cast<const char*>(tfpp_zero_page);
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_tz[i] ? zp : a_base + a_vo[i]), (lds_void_t*)(stage + (i * NWAVES + wave_u) * 1024u), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < B_INST; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_tz[j] ? zp : b_base + b_vo[j]), (lds_void_t*)(stage + A_BYTES + (j * NWAVES + wave_u) * 1024u), 16, 0, 0);
      }
    };
    const bool ptail = tail < BKP;  // wave-uniform
    unsigned wr = lds_base, rd = lds_base;
#pragma unroll
    for (int s2 = 0; s2 < NSTAGE - 1; ++s2)
      if (s2 < nst_u) { issue(wr, ptail && s2 == nst_u - 1); wr += STAGE_BYTES; }
    int st = 0;
#pragma unroll 1
    for (; st + NSTAGE - 1 < nst_u; ++st) {  // steady state: NSTAGE - 2 younger stages stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LOADS) : "memory");
      __builtin_amdgcn_s_barrier();  // stage st landed for every wave; every wave is done with stage st - 1, whose slot is refilled now
      issue(wr, ptail && st + NSTAGE - 1 == nst_u - 1);
      wr += STAGE_BYTES;
      if (wr == ring_end) wr = lds_base;
      stage_mma(rd);
      rd += STAGE_BYTES;
      if (rd == ring_end) rd = lds_base;
    }
#pragma unroll 1
    for (; st < nst_u; ++st) {  // drain
      switch (nst_u - 1 - st) {  // wave-uniform, <= NSTAGE - 2
#define TFPP_WAIT_CASE(A) case A: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * LOADS) : "memory"); break
        TFPP_WAIT_CASE(1); TFPP_WAIT_CASE(2); TFPP_WAIT_CASE(3); TFPP_WAIT_CASE(4); TFPP_WAIT_CASE(5); TFPP_WAIT_CASE(6);
#undef TFPP_WAIT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      stage_mma(rd);
      rd += STAGE_BYTES;
      if (rd == ring_end) rd = lds_base;
    }
  } else {
    // this thread's chunks of every stage: DMA i of the workgroup fills bytes [i*NT*16, (i+1)*NT*16) of the stage half, thread tid the
    // 16 bytes at slot q = i*NT + tid -> pixel row pk = q / CH, physical chunk cp = q % CH, logical chunk cp ^ swz(pk).
    // Addressing is incremental: every chunk keeps a running source pointer that advances by one stage (BKP pixels) per issue -- two
    // VALU adds per DMA instead of a 64-bit multiply-add plus range checks (the first version spent ~130 VALU and ~150 SALU
    // instructions per stage there, against 16 MFMAs: SQ_INSTS_VALU / SQ_INSTS_MFMA = 8.1).  Chunks outside the tile (channel >= n_g,
    // column >= KK) point at the zero page with step 0; pixels past the end of the slice exist only in the last stage and are
    // masked there.  3x3 / strided layers (no constant stride between stages) keep per-stage address generation, in 32-bit arithmetic.
    const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);
    int a_pk[A_INST];
    const T* a_cur[A_INST];
    long a_step[A_INST];
  #pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int q = i * NT + tid;
      a_pk[i] = q / CH_A;
      const int n = bm0 + (((q % CH_A) ^ swz<TM>(a_pk[i])) * 8);  // dY channel of the chunk
      const bool ok = n < p.n_g;
      a_cur[i] = ok ? dy + (size_t)(p_beg + a_pk[i]) * p.dy_ld + n : zero;
      a_step[i] = ok ? (long)BKP * p.dy_ld : 0;
    }
    int b_pk[B_INST], b_c[B_INST], b_r[B_INST], b_s[B_INST];
    bool b_ok[B_INST];
    const T* b_cur[B_INST];
    long b_step[B_INST];
  #pragma unroll
    for (int j = 0; j < B_INST; ++j) {
      const int q = j * NT + tid;
      b_pk[j] = q / CH_B;
      const int kk = bn0 + (((q % CH_B) ^ swz<TN>(b_pk[j])) * 8);  // gathered column (tap, channel) of the chunk
      b_ok[j] = kk < KK;
      const int rs = b_ok[j] ? kk / p.ks_g : 0;
      b_c[j] = kk - rs * p.ks_g;
      b_r[j] = rs / p.S;
      b_s[j] = rs - b_r[j] * p.S;
      const bool lin = pointwise && b_ok[j];
      b_cur[j] = lin ? x + (size_t)(p_beg + b_pk[j]) * p.x_ld + b_c[j] : zero;
      b_step[j] = lin ? (long)BKP * p.x_ld : 0;
    }

    auto issue = [&](int st) {  // LDS-DMA of pixel stage st into ring slot st % NSTAGE; called with st = 0, 1, 2, ... in order
      const unsigned stage = lds_base + (unsigned)((st % NSTAGE) * STAGE_BYTES);
      const bool last = (st == nst - 1) && tail < BKP;  // wave-uniform
  #pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        const T* ga = a_cur[i];
        if (last && a_pk[i] >= tail) ga = zero;
        a_cur[i] += a_step[i];
        __builtin_amdgcn_global_load_lds((gbl_void_t*)ga, (lds_void_t*)(stage + (unsigned)((i * NWAVES + wave) * 1024)), 16, 0, 0);
      }
  #pragma unroll
      for (int j = 0; j < B_INST; ++j) {
        const T* gb = b_cur[j];
        if (pointwise) {
          if (last && b_pk[j] >= tail) gb = zero;
          b_cur[j] += b_step[j];
        } else {
          gb = zero;
          const int pl = st * BKP + b_pk[j];  // pixel inside the slice
          if (pl < npix && b_ok[j]) {
            const unsigned pix = (unsigned)(p_beg + pl);  // P < 2^31 (checked by the dispatcher)
            const unsigned b = pix / (unsigned)hw, rem = pix - b * (unsigned)hw, hd = rem / (unsigned)p.Wd, wd = rem - hd * (unsigned)p.Wd;
            const int hs = (int)hd * p.stride - p.pad + b_r[j], ws = (int)wd * p.stride - p.pad + b_s[j];
            if (hs >= 0 && hs < p.Hs && ws >= 0 && ws < p.Ws) gb = x + ((size_t)((int)b * p.Hs + hs) * p.Ws + ws) * p.x_ld + b_c[j];
          }
        }
        __builtin_amdgcn_global_load_lds((gbl_void_t*)gb, (lds_void_t*)(stage + (unsigned)(A_BYTES + (j * NWAVES + wave) * 1024)), 16, 0, 0);
      }
    };

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
      stage_mma(lds_base + (unsigned)((st % NSTAGE) * STAGE_BYTES));
    }
  }

  // ---- epilogue: slice -> workspace [split][G*n_g][KK] (summed by wgrad_reduce_kernel), or straight into dw
  const int RS = p.R * p.S;
  if (p.ws && p.splits > 1) {
    float* __restrict__ wsp = p.ws + ((size_t)split * p.G * p.n_g + (size_t)g * p.n_g) * KK;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = bm0 + wm * WM + i * 16 + kg * 4 + r;
        if (n >= p.n_g) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int kk = bn0 + wn * WN + j * 16 + m16;
          if (kk < KK) wsp[(size_t)n * KK + kk] = acc[i][j][r];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = bm0 + wm * WM + i * 16 + kg * 4 + r;
      if (n >= p.n_g) continue;
      int row = g * p.n_g + n;
      if (p.row_map) row = p.row_map[row];
      if (row < 0) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int kk = bn0 + wn * WN + j * 16 + m16;
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

template <int TM, int TN, int WGM, int WGN, int BKP, int NSTAGE, bool PW = false>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_wgrad_glds_kernel(tfpp_wgrad_params p) {
  wgrad_glds_tile<TM, TN, WGM, WGN, BKP, NSTAGE, PW>(p, (int)blockIdx.x);
}

// Grouped launch (round 5): the pointwise weight gradients of MANY layers in one grid.  A flush of the weight-gradient lane used to be
// ~100 launches of this kernel plus ~100 slice sums, each a few dozen workgroups, one after the other on a stream: neither the chip nor
// the stream was ever full.  Here the workgroups of all layers of a batch form one grid (the descriptor table travels in the kernel
// arguments, so a captured hipGraph node carries it), the pixel reduction of a layer is split only as far as the WHOLE grid needs it,
// and a grid smaller than the work list (persistent workgroups, gridDim.x < grp.total) caps the CUs the lane takes from the dY chain.
template <int TM, int TN, int WGM, int WGN, int BKP, int NSTAGE>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_wgrad_glds_group_kernel(const tfpp_wgrad_group grp) {
  for (int id = (int)blockIdx.x; id < grp.total; id += (int)gridDim.x) {
    int k = 0;
    for (int i = 1; i < grp.n; ++i) k = (id >= grp.it[i].wg_start) ? i : k;  // wave-uniform (kernel arguments, blockIdx)
    k = __builtin_amdgcn_readfirstlane(k);
    const int local = id - grp.it[k].wg_start;
    if (local < grp.it[k].wgs) {  // (the workgroup ranges are padded to whole XCD rounds)
      const tfpp_wgrad_params p = tfpp_wgrad_item_params(grp.it[k]);
      wgrad_glds_tile<TM, TN, WGM, WGN, BKP, NSTAGE, true>(p, local, grp.it[k].pin);
    }
    if (id + (int)gridDim.x < grp.total) {  // another tile follows: every wave is done with the LDS ring before its first DMA, and the
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // epilogue's loads / stores no longer count against the ring's vmcnt arithmetic
      __syncthreads();
    }
  }
}

template <int TM, int TN, int WGM, int WGN, int BKP, int NSTAGE> int launch_wgrad_glds(const tfpp_wgrad_params& p, hipStream_t st) {
  const int KK = p.R * p.S * p.ks_g;
  // TFPP_WGRAD_MIN_LDS (bytes): occupancy limiter for A/B runs -- weight gradients run beside the latency-bound dY chain of the other
  // streams; a larger allocation leaves fewer of their workgroups per CU and so more wave slots for that chain.
  static const size_t min_lds = [] { const char* e = std::getenv("TFPP_WGRAD_MIN_LDS"); return e ? (size_t)std::atol(e) : (size_t)0; }();
  constexpr size_t need = (size_t)NSTAGE * BKP * (TM + TN) * 2;
  const size_t lds = need > min_lds ? need : min_lds;
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_glds_kernel<TM, TN, WGM, WGN, BKP, NSTAGE, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_glds_kernel<TM, TN, WGM, WGN, BKP, NSTAGE, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  dim3 grid((unsigned)((long)p.G * p.splits * cdiv(p.n_g, TM) * cdiv(KK, TN)));
  // lean loop: pointwise layer whose operand byte offsets fit 32 bits (TFPP_WGRAD_PW=0: the general loop, for A/B runs)
  static const int pw_env = [] { const char* e = std::getenv("TFPP_WGRAD_PW"); return e ? std::atoi(e) : 1; }();
  const long P = (long)p.B * p.Hd * p.Wd;
  const bool pw = pw_env && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd && p.n_g >= 8 && KK >= 8 &&
                  (P + 2 * BKP) * (long)p.dy_ld * 2 < (1l << 32) && (P + 2 * BKP) * (long)p.x_ld * 2 < (1l << 32);
  if (pw) hipLaunchKernelGGL((conv_wgrad_glds_kernel<TM, TN, WGM, WGN, BKP, NSTAGE, true>), grid, dim3(WGM * WGN * 64), lds, st, p);
  else hipLaunchKernelGGL((conv_wgrad_glds_kernel<TM, TN, WGM, WGN, BKP, NSTAGE, false>), grid, dim3(WGM * WGN * 64), lds, st, p);
  TFPP_CHECK_LAUNCH();
  return 0;
}
}  // namespace

bool wgrad_glds_supported(const tfpp_wgrad_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_WGRAD_GLDS"); return (e && e[0] == '0') ? 0 : 1; }();
  const int KK = p.R * p.S * p.ks_g;
  return on && dtype == TFPP_BF16 && p.n_g > 32 && KK > 32 && p.n_g % 8 == 0 && p.ks_g % 8 == 0 && p.dy_ld % 8 == 0 && p.x_ld % 8 == 0 &&
         ((uintptr_t)p.dy & 15) == 0 && ((uintptr_t)p.x & 15) == 0;
}

// 128 x 128 tiles pay when both dimensions fill them reasonably (>= 75 % of the padded tile area is real work)
bool wgrad_glds128_preferred(const tfpp_wgrad_params& p) {
  static const int on = [] { const char* e = std::getenv("TFPP_WGRAD_GLDS128"); return (e && e[0] == '0') ? 0 : 1; }();
  const int KK = p.R * p.S * p.ks_g;
  if (!on || p.n_g < 128 || KK < 128) return false;
  const double fill = ((double)p.n_g * KK) / ((double)cdiv(p.n_g, 128) * 128 * cdiv(KK, 128) * 128);
  return fill >= 0.75;
}

int conv_wgrad_glds(const tfpp_wgrad_params& p, int tile, hipStream_t st) {
  static const int cfg = [] { const char* e = std::getenv("TFPP_WGRAD_CFG"); return e ? std::atoi(e) : 0; }();
  if (tile == 128) {
    // two rings of 2 x 32 KB share a CU when the launch has at least ~2 workgroups per CU (fusion MLP 3840x6048x1512: 561 vs 477 TFLOP/s);
    // with fewer workgroups the deeper 3 x 32 KB ring of a lone workgroup wins (3840x1512x1512, 144 workgroups: 331 vs 281)
    const long wgs = (long)p.G * p.splits * cdiv(p.n_g, 128) * cdiv(p.R * p.S * p.ks_g, 128);
    if (cfg == 1 || (cfg == 0 && wgs >= 400)) return launch_wgrad_glds<128, 128, 2, 4, 64, 2>(p, st);
    return launch_wgrad_glds<128, 128, 2, 4, 64, 3>(p, st);
  }
  if (cfg == 2) return launch_wgrad_glds<64, 64, 2, 2, 64, 3>(p, st);  // 64-pixel stages
  return launch_wgrad_glds<64, 64, 2, 2, 32, 4>(p, st);
}

// ---- grouped launch (host side) --------------------------------------------------------------------------------------------------
// A layer can join a group when the lean pointwise loop covers it: 1x1 / stride 1 / one group, 32-bit operand byte offsets.
bool wgrad_glds_group_ok(const tfpp_wgrad_params& p, int dtype) {
  static const int on = [] { const char* e = std::getenv("TFPP_WGRAD_GROUP"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || !wgrad_glds_supported(p, dtype)) return false;
  const long P = (long)p.B * p.Hd * p.Wd;
  return p.G == 1 && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd && p.n_g > 32 && p.ks_g > 32 &&
         P < (1l << 30) && (P + 128) * (long)p.dy_ld * 2 < (1l << 32) && (P + 128) * (long)p.x_ld * 2 < (1l << 32) && p.dw_ld < (1l << 31);
}

template <int TM, int TN, int WGM, int WGN, int BKP, int NSTAGE> static int launch_wgrad_glds_group(const tfpp_wgrad_group& grp, int grid_cap, hipStream_t st) {
  constexpr size_t need = (size_t)NSTAGE * BKP * (TM + TN) * 2;
  // TFPP_WGRAD_GROUP_LDS (bytes): occupancy limiter of the grouped grids, for A/B runs (see launch_wgrad_glds)
  static const size_t min_lds = [] { const char* e = std::getenv("TFPP_WGRAD_GROUP_LDS"); return e ? (size_t)std::atol(e) : (size_t)0; }();
  const size_t lds = (TM == 128 && min_lds > need) ? min_lds : need;
  static_assert(need <= 160 * 1024, "ring must fit the LDS of a CU");
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask))
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_glds_group_kernel<TM, TN, WGM, WGN, BKP, NSTAGE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int grid = grp.total;
  if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
  hipLaunchKernelGGL((conv_wgrad_glds_group_kernel<TM, TN, WGM, WGN, BKP, NSTAGE>), dim3((unsigned)grid), dim3(WGM * WGN * 64), lds, st, grp);
  TFPP_CHECK_LAUNCH();
  return 0;
}

int conv_wgrad_glds_group(const tfpp_wgrad_group& grp, int tile, int grid_cap, hipStream_t st) {
  if (grp.n < 1 || grp.total < 1) return 0;
  // 128 x 128: the 2 x 32 KB ring (two workgroups per CU, or one beside a forward / data-gradient GEMM of the dY chain)
  if (tile == 128) return launch_wgrad_glds_group<128, 128, 2, 4, 64, 2>(grp, grid_cap, st);
  // 256 x 256 (round 6; layers with n_g, KK >= 1024: the C = 1512 fusion transformer and stage 4): 8 waves of 128 x 64, 32-pixel stages,
  // 4 x 32 KB ring, one workgroup per CU.  Half the operand bytes per FLOP of the 128 x 128 tile, and a 3840 x 6048 x 1512 layer is 144
  // workgroups -- ONE round of the chip, so the tiles that share a panel start together and meet in the L2.
  if (tile == 256) return launch_wgrad_glds_group<256, 256, 2, 4, 32, 4>(grp, grid_cap, st);
  return launch_wgrad_glds_group<64, 64, 2, 2, 32, 4>(grp, grid_cap, st);
}
