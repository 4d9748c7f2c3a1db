// "Ping-pong" LDS-DMA GEMM for the large pointwise bf16 layers (fusion-transformer linears transfuser.py:352-359,383-402 and the
// stage-4 1x1 convolutions): C[M][N] = epilogue(A[M][K] * W[N][K]^T), forward and data gradient.
//
// Why another kernel: the ring kernel of gemm_glds.hip runs all waves of a workgroup in lock step (read fragments -> MFMA -> barrier), so the
// matrix pipe idles while fragments are read and while the barrier drains; it tops out near 800 TFLOP/s on 3840x6048x1512.  Here the 8 waves
// of a workgroup form two groups (waves 0-3 / 4-7; every SIMD hosts one wave of each) that run the SAME phase sequence one barrier apart:
//
//     group 0:  L0 | C0 | L1 | C1 | ...          L = fragment reads (ds_read_b128) + LDS-DMA issue + counted waits
//     group 1:     | L0 | C0 | L1 | C1 | ...     C = MFMAs only (s_setprio 1)
//
// so in every interval between two barriers one wave of each SIMD feeds the matrix pipe while the other one loads
// (cdna_hip_programming.md 5.5 T3+T4/T5: phase-split schedule, counted vmcnt, priority around the MFMA cluster).
//
// Tile BM x BN x 64 (BM = 256 / 128, BN = 256 / 192 / 128), wave grid 2 (M) x 4 (N): a wave owns WM = BM/2 rows and WN = BN/4 columns.  A K tile
// is consumed in NPH = WM/32 phases: phase p multiplies the wave's 32-row slab p of A with ALL its B fragments (read once, in phase 0) --
// every operand byte is read from LDS exactly once per wave.  Per wave and phase: 16 MFMAs 16x16x32 or 8 MFMAs 32x32x16 at BN = 256.
//
// LDS: two K-tile buffers [B: BN rows][A: BM rows] of 128-byte rows (64 bf16), filled by global_load_lds_dwordx4 in 1 KB pieces (8 rows);
// wave w issues pieces w, w + 8, ... of either operand.  The A image is stored slab-major (LDS row j = slab j/64, group (j/32)&1, row j%32) so
// that the slab read in phase p is dead after phase p and can be refilled by the K tile after next while the rest of the buffer is still in
// use: a piece is re-issued one phase after its last reader, i.e. 5..7 phases (1.25..1.75 K tiles) before it is needed, and every L phase ends
// with `s_waitcnt vmcnt(LPW)` (LPW = loads per wave and K tile): everything older than one K tile of loads has landed.  Hazards:
//   RAW  a piece is read (by either group) at the earliest one barrier after BOTH groups executed the vmcnt that covers it;
//   WAR  an L phase ends with lgkmcnt(0) BEFORE its barrier, so a buffer region read in phase j by group 0 (interval 2j) and group 1
//        (interval 2j + 1) is free from interval 2j + 2 on, which is group 0's phase j + 1.
// Bank conflicts: physical 16-byte slot of logical chunk c in LDS row r is c ^ ((r >> 1) & 7), applied on the DMA source address and on the
// fragment reads (rule 21); conflict-free for the lane groups of ds_read_b128 with both the 16x16x32 and the 32x32x16 fragment shapes.
//
// Workgroup -> tile map: the tile grid is cut into XM x XN = 8 rectangles, one per XCD (block b runs on XCD b % 8), chosen to minimise
// XN * |A| + XM * |B| -- each activation panel crosses the fabric XN times and each weight panel XM times instead of up to 8.
#include "gemm_core.cuh"
#include "gemm_internal.h"
#include <cstdlib>
#include <utility>

typedef __attribute__((address_space(3))) void pp_lds_void_t;
typedef const __attribute__((address_space(1))) void pp_gbl_void_t;
typedef unsigned int pp_u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __attribute__((aligned(16))) unsigned int tfpp_pp_zero_page[4] = {0u, 0u, 0u, 0u};

struct PPMap { int tm, tn, xm, xn, dbg; };  // dbg: ablation bits (tools/gemm_pp_micro.py): 1 no steady-state DMA, 2 no fragment reads, 4 no MFMAs, 8 no stores

template <int OFF> __device__ __forceinline__ uint4 pp_ds_read(unsigned addr) {  // OFF: 16-bit immediate byte offset
  pp_u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

template <typename F, int... I> __device__ __forceinline__ void pp_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void pp_static_for(F&& f) {
  pp_static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int N> __device__ __forceinline__ void pp_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, bool MF32>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(tfpp_conv_params p, PPMap map) {
  constexpr int WM = BM / 2, WN = BN / 4, NPH = WM / 32, NB = BN / 64, NA = BM / 64, LPW = NB + NA;
  constexpr int KT_BYTES = (BM + BN) * 128, A_OFF = BN * 128;
  constexpr int FN = WN / 16, FN32 = WN / 32, NKS = MF32 ? 4 : 2;
  constexpr int LATE = NPH == 4 ? 2 : 1;  // A loads of K tile t + 1 issued in phase 0 of tile t
  static_assert(NPH == 4 || NPH == 2, "BM = 256 or 128");
  static_assert(NA == NPH && WN % 16 == 0 && (!MF32 || WN % 32 == 0), "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- tile of this workgroup (see the header): XCD x = blockIdx.x % 8 owns rectangle (x / xn, x % xn) of the tile grid
  int mtile, ntile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int xi = xcd / map.xn, xj = xcd - xi * map.xn;
    if (xi >= map.xm) return;  // fewer than 8 rectangles (tiny grids)
    const int qm = map.tm / map.xm, em = map.tm - qm * map.xm, qn = map.tn / map.xn, en = map.tn - qn * map.xn;
    const int rm = qm + (xi < em ? 1 : 0), m0 = xi * qm + (xi < em ? xi : em);
    const int rn = qn + (xj < en ? 1 : 0), n0 = xj * qn + (xj < en ? xj : en);
    if (idx >= rm * rn) return;  // uniform: before any barrier
    const int in = idx / rm;
    mtile = __builtin_amdgcn_readfirstlane(m0 + idx - in * rm);
    ntile = __builtin_amdgcn_readfirstlane(n0 + in);
  }
  const int split = __builtin_amdgcn_readfirstlane((int)blockIdx.y);
  const int M = p.B * p.Hd * p.Wd, K = p.ks_g;
  const int bm0 = mtile * BM, bn0 = ntile * BN;
  const unsigned lds_base = (unsigned)(size_t)(pp_lds_void_t*)smem;

  // ---- K tiles of this workgroup
  int kt_beg = 0, nkt = (K + 63) / 64;
  if (p.splitk > 1) {
    const int per = (nkt + p.splitk - 1) / p.splitk;
    kt_beg = split * per;
    nkt = (kt_beg + per < nkt ? kt_beg + per : nkt) - kt_beg;
    if (nkt < 0) nkt = 0;
  }
  kt_beg = __builtin_amdgcn_readfirstlane(kt_beg);
  nkt = __builtin_amdgcn_readfirstlane(nkt);
  const bool ktail = (kt_beg + nkt) * 64 > K;  // the last K tile of this workgroup reaches past K

  // ---- LDS-DMA bookkeeping: every piece of this lane fetches logical chunk kc of its row (the swizzle depends on the piece's parity only)
  const int kc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const bool tz = ktail && ((kt_beg + nkt - 1) * 64 + kc * 8 >= K);  // this lane's chunk of the LAST tile lies beyond K: zero page
  unsigned b_vo[NB], a_vo[NA];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int row = (wave + 8 * k) * 8 + (lane >> 3);
    const int n = bn0 + row < p.n_g ? bn0 + row : p.n_g - 1;  // rows past the end repeat the last one: finite data, never stored
    b_vo[k] = (unsigned)n * (unsigned)K * 2u + (unsigned)kc * 16u;
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const int trow = (wave >> 2) * WM + k * 32 + (wave & 3) * 8 + (lane >> 3);  // LDS A row (wave + 8k) * 8 + (lane >> 3), slab-major
    const int m = bm0 + trow < M ? bm0 + trow : M - 1;
    a_vo[k] = (unsigned)m * (unsigned)p.src_ld * 2u + (unsigned)kc * 16u;
  }
  const char* a_base = reinterpret_cast<const char*>(p.src) + (size_t)kt_beg * 128;  // uniform
  const char* b_base = reinterpret_cast<const char*>(p.w) + (size_t)kt_beg * 128;
  const char* zero = reinterpret_cast<const char*>(tfpp_pp_zero_page);
  const unsigned wave_u = (unsigned)wave;
  // one load: piece k of K tile tt (index within this workgroup's K range) into buffer buf; `last` (wave-uniform): tt is the tile with the K
  // tail, whose chunks beyond K come from the zero page (both operands: NaN * 0 must not happen)
  auto issue_b = [&](int k, int tt, unsigned buf, bool last) {
    const char* base = b_base + (size_t)tt * 128;  // uniform
    pp_lds_void_t* dst = (pp_lds_void_t*)(size_t)(lds_base + buf * KT_BYTES + (wave_u + 8u * k) * 1024u);
    if (!last) __builtin_amdgcn_global_load_lds((pp_gbl_void_t*)(base + b_vo[k]), dst, 16, 0, 0);
    else {
      asm volatile("; K tail" ::: "memory");  // keeps the two paths apart: merged, every load of the steady state pays the address select
      __builtin_amdgcn_global_load_lds((pp_gbl_void_t*)(tz ? zero : base + b_vo[k]), dst, 16, 0, 0);
    }
  };
  auto issue_a = [&](int k, int tt, unsigned buf, bool last) {
    const char* base = a_base + (size_t)tt * 128;
    pp_lds_void_t* dst = (pp_lds_void_t*)(size_t)(lds_base + buf * KT_BYTES + A_OFF + (wave_u + 8u * k) * 1024u);
    if (!last) __builtin_amdgcn_global_load_lds((pp_gbl_void_t*)(base + a_vo[k]), dst, 16, 0, 0);
    else {
      asm volatile("; K tail" ::: "memory");
      __builtin_amdgcn_global_load_lds((pp_gbl_void_t*)(tz ? zero : base + a_vo[k]), dst, 16, 0, 0);
    }
  };

  // ---- fragment read bases (per K step; the swizzle term depends on the lane only: all row offsets are multiples of 16 / 32)
  unsigned a_rd[2][NKS], b_rd[2][NKS];
  if constexpr (!MF32) {
    const int r16 = lane & 15, kg = lane >> 4, s = r16 >> 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned phys = (unsigned)((ks * 4 + kg) ^ s) * 16u;
      a_rd[0][ks] = lds_base + A_OFF + (unsigned)(wm * 32 + r16) * 128u + phys;
      b_rd[0][ks] = lds_base + (unsigned)(wn * WN + r16) * 128u + phys;
      a_rd[1][ks] = a_rd[0][ks] + KT_BYTES;
      b_rd[1][ks] = b_rd[0][ks] + KT_BYTES;
    }
  } else {
    const int r32 = lane & 31, kg = lane >> 5, s = (r32 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned phys = (unsigned)((ks * 2 + kg) ^ s) * 16u;
      a_rd[0][ks] = lds_base + A_OFF + (unsigned)(wm * 32 + r32) * 128u + phys;
      b_rd[0][ks] = lds_base + (unsigned)(wn * WN + r32) * 128u + phys;
      a_rd[1][ks] = a_rd[0][ks] + KT_BYTES;
      b_rd[1][ks] = b_rd[0][ks] + KT_BYTES;
    }
  }

  // ---- accumulators and fragments (16x16x32: acc[slab * 2 + i][j]; 32x32x16: acc32[slab][j])
  f32x4_t acc[MF32 ? 1 : 2 * NPH][MF32 ? 1 : FN];
  f32x16_t acc32[MF32 ? NPH : 1][MF32 ? FN32 : 1];
  if constexpr (!MF32) {
#pragma unroll
    for (int i = 0; i < 2 * NPH; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < NPH; ++i)
#pragma unroll
      for (int j = 0; j < FN32; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  }
  Frag<bf16_t> fa[NKS][MF32 ? 1 : 2], fb[NKS][MF32 ? FN32 : FN];
  const int dbg = map.dbg;
  if (dbg & 2) {  // timing experiments without fragment reads multiply zeros
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int i = 0; i < (MF32 ? 1 : 2); ++i) fa[ks][i].v = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < (MF32 ? FN32 : FN); ++j) fb[ks][j].v = make_uint4(0, 0, 0, 0);
    }
  }

  // one K tile: t = index within this workgroup's K range, in LDS buffer BUF; mode 0: tiles t + 1 and t + 2 exist, 1: t + 1 is the last, 2: t is
  auto tile = [&](auto BUF_, int t, int mode) {
    constexpr int BUF = decltype(BUF_)::value;
    pp_static_for<NPH>([&](auto P_) {
      constexpr int P = decltype(P_)::value;
      // ================= L phase: fragment reads
      if (dbg & 2) {
      } else if constexpr (!MF32) {
        if constexpr (P == 0) {
          pp_static_for<2>([&](auto KS_) {
            constexpr int KS = decltype(KS_)::value;
            pp_static_for<FN>([&](auto J_) {
              constexpr int J = decltype(J_)::value;
              fb[KS][J].v = pp_ds_read<J * 16 * 128>(b_rd[BUF][KS]);
            });
          });
        }
        pp_static_for<2>([&](auto KS_) {
          constexpr int KS = decltype(KS_)::value;
          fa[KS][0].v = pp_ds_read<(P * 64) * 128>(a_rd[BUF][KS]);
          fa[KS][1].v = pp_ds_read<(P * 64 + 16) * 128>(a_rd[BUF][KS]);
        });
      } else {
        if constexpr (P == 0) {
          pp_static_for<4>([&](auto KS_) {
            constexpr int KS = decltype(KS_)::value;
            pp_static_for<FN32>([&](auto J_) {
              constexpr int J = decltype(J_)::value;
              fb[KS][J].v = pp_ds_read<J * 32 * 128>(b_rd[BUF][KS]);
            });
          });
        }
        pp_static_for<4>([&](auto KS_) {
          constexpr int KS = decltype(KS_)::value;
          fa[KS][0].v = pp_ds_read<(P * 64) * 128>(a_rd[BUF][KS]);
        });
      }
      // ================= L phase: LDS-DMA of the pieces whose previous contents were read one phase ago
      if constexpr (P == 0) {  // late A slabs of K tile t + 1 (other buffer; their slots were read in the last phase(s) of tile t - 1)
        if (mode != 2 && !(dbg & 1)) {
          const bool last = ktail && (t + 1 == nkt - 1);
#pragma unroll
          for (int k = NA - LATE; k < NA; ++k) issue_a(k, t + 1, BUF ^ 1, last);
        }
      } else if (mode == 0 && !(dbg & 1)) {  // K tile t + 2 into this buffer
        const bool last = ktail && (t + 2 == nkt - 1);
        if constexpr (NPH == 4) {
          if constexpr (P == 1) {
#pragma unroll
            for (int k = 0; k < (NB < 2 ? NB : 2); ++k) issue_b(k, t + 2, BUF, last);
          } else if constexpr (P == 2) {
#pragma unroll
            for (int k = 2; k < NB; ++k) issue_b(k, t + 2, BUF, last);
          } else {
            issue_a(0, t + 2, BUF, last);
            issue_a(1, t + 2, BUF, last);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NB; ++k) issue_b(k, t + 2, BUF, last);
          issue_a(0, t + 2, BUF, last);
        }
      }
      // ================= L phase: everything older than one K tile of loads has landed (see the header); my reads have returned
      if (mode == 0) pp_wait_vmcnt<LPW>();
      else if (mode == 1) {
        constexpr int NB1 = NB < 2 ? NB : 2;
        constexpr int W = NPH == 4 ? (P == 0 ? LPW : P == 1 ? LPW - NB1 : P == 2 ? LPW - NB : LPW - NB - 2) : (P == 0 ? LPW : 1);
        pp_wait_vmcnt<W>();
      } else if (P == 0) pp_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ================= C phase
      __builtin_amdgcn_s_setprio(1);
      if (dbg & 4) {
      } else if constexpr (!MF32) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) frag_mma(fa[ks][i], fb[ks][j], acc[P * 2 + i][j]);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < FN32; ++j)
            acc32[P][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[ks][0].v), __builtin_bit_cast(bf16x8_t, fb[ks][j].v),
                                                                  acc32[P][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  if (nkt > 0) {
    // ---- prologue: K tile 0 completely, K tile 1 without its late A slabs (phase 0 of tile 0 issues those)
    {
      const bool last0 = ktail && nkt == 1, last1 = ktail && nkt == 2;
#pragma unroll
      for (int k = 0; k < NB; ++k) issue_b(k, 0, 0, last0);
#pragma unroll
      for (int k = 0; k < NA; ++k) issue_a(k, 0, 0, last0);
      if (nkt > 1) {
#pragma unroll
        for (int k = 0; k < NB; ++k) issue_b(k, 1, 1, last1);
#pragma unroll
        for (int k = 0; k < NA - LATE; ++k) issue_a(k, 1, 1, last1);
        pp_wait_vmcnt<LPW - LATE>();
      } else {
        pp_wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (wm == 1) {  // group 1 runs one interval behind group 0
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int t = 0; t < nkt; t += 2) {
      const int r0 = nkt - 1 - t;  // tiles after t
      tile(std::integral_constant<int, 0>{}, t, r0 >= 2 ? 0 : (r0 == 1 ? 1 : 2));
      if (r0 >= 1) tile(std::integral_constant<int, 1>{}, t + 1, r0 >= 3 ? 0 : (r0 == 2 ? 1 : 2));
    }
    if (wm == 0) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (dbg & 8) {  // no stores: keep the accumulators alive behind a condition no data satisfies
    float sum = 0.f;
    if constexpr (!MF32) {
#pragma unroll
      for (int i = 0; i < 2 * NPH; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
#pragma unroll
      for (int i = 0; i < NPH; ++i)
#pragma unroll
        for (int j = 0; j < FN32; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += acc32[i][j][e];
    }
    if (sum != 1.2345678e33f) return;
  }

  // ---- split-K: raw fp32 slice -> workspace [split][M][n_g]; splitk_epilogue_kernel finishes the job
  if (p.splitk > 1) {
    float* __restrict__ wsp = p.splitk_ws + (size_t)split * M * p.n_g;
    if constexpr (!MF32) {
#pragma unroll
      for (int i = 0; i < 2 * NPH; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
          if (m >= M) continue;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const int n = bn0 + wn * WN + j * 16 + (lane & 15);
            if (n < p.n_g) wsp[(size_t)m * p.n_g + n] = acc[i][j][r];
          }
        }
    } else {
#pragma unroll
      for (int i = 0; i < NPH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = bm0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m >= M) continue;
#pragma unroll
          for (int j = 0; j < FN32; ++j) {
            const int n = bn0 + wn * WN + j * 32 + (lane & 31);
            if (n < p.n_g) wsp[(size_t)m * p.n_g + n] = acc32[i][j][r];
          }
        }
    }
    return;
  }

  // ---- epilogue: 16-row passes through a per-wave LDS strip (the ring is dead: every DMA was waited for, every read has returned)
  __syncthreads();
  float* strip = reinterpret_cast<float*>(smem) + wave * EpiStrip<FN>::FLOATS;
  if constexpr (!MF32) {
#pragma unroll
    for (int i = 0; i < 2 * NPH; ++i) {
      const int m_pass = bm0 + wm * WM + i * 16;
      epi_pass_bf16<FN, 2 * NPH, false>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, 0);
    }
  } else {
    constexpr int PITCH = EpiStrip<FN>::PITCH;
    const int c32 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < NPH; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < FN32; ++j)
#pragma unroll
          for (int r = 0; r < 8; ++r) strip[((r & 3) + 8 * (r >> 2) + 4 * hi) * PITCH + j * 32 + c32] = acc32[i][j][8 * h + r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int m_pass = bm0 + wm * WM + i * 32 + h * 16;
        epi_finish_bf16<FN>(p, strip, lane, m_pass, M - m_pass, bn0 + wn * WN, 0);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// configurations: index -> (BM, BN, 32x32x16 MFMA)
struct PPCfg { int bm, bn, mf32; };
static const PPCfg kPPCfg[] = {{256, 256, 0}, {256, 256, 1}, {256, 192, 0}, {256, 128, 0}, {256, 128, 1},
                               {128, 256, 0}, {128, 256, 1}, {128, 192, 0}, {128, 128, 0}, {128, 128, 1}};
static constexpr int kPPNumCfg = (int)(sizeof(kPPCfg) / sizeof(kPPCfg[0]));

// tuning hook (tfpp_gemm_pp_config): 0 = automatic choice, -1 = kernel off, 1 + i + 100 * splits = configuration i with `splits` K slices (0: automatic)
static int g_pp_force = [] { const char* e = std::getenv("TFPP_GEMM_PP"); return e ? std::atoi(e) : 0; }();
static int g_pp_dbg = 0;  // ablation bits (timing experiments only: results are wrong with any bit set), cfg / 10000
extern "C" int tfpp_gemm_pp_config(int cfg) {
  g_pp_dbg = cfg >= 10000 ? cfg / 10000 : 0;
  g_pp_force = cfg >= 10000 ? cfg % 10000 : cfg;
  return 0;
}

bool conv_pp_supported(const tfpp_conv_params& p, int dtype) {
  if (g_pp_force < 0 || dtype != TFPP_BF16) return false;
  const long M = (long)p.B * p.Hd * p.Wd;
  const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd && p.G == 1;
  if (!pointwise || p.stats_partial || p.bns_partial || p.dst_nchw || p.dst_f32) return false;
  if (((p.n_g | (int)p.dst_ld | p.ks_g | (int)p.src_ld) & 7) || ((uintptr_t)p.dst & 15) || ((uintptr_t)p.src & 15) || ((uintptr_t)p.w & 15)) return false;
  if (p.res && ((((int)p.res_ld) & 7) || ((uintptr_t)p.res & 15))) return false;
  if (M * (long)p.src_ld * 2 >= (1l << 32) || (long)p.n_g * p.ks_g * 2 >= (1l << 32)) return false;  // 32-bit DMA offsets
  if (g_pp_force > 0) return M >= 128 && p.n_g >= 128 && p.ks_g >= 64;
  return M >= 2048 && p.n_g >= 1024 && p.ks_g >= 1024;  // the fusion-transformer / stage-4 shapes
}

// number of launch rounds (in units of one 256-CU round of this tile shape, weighted by the tile's work) for a plan
static double pp_cost(long M, long N, long K, const PPCfg& c, int splits) {
  const long tiles = (long)cdiv(M, c.bm) * cdiv(N, c.bn) * splits;
  const long rounds = (tiles + 255) / 256;
  const double kt = (double)cdiv(cdiv(K, 64), splits);
  // per tile: K loop (one unit per 64 x 256 x 256 MACs; smaller tiles feed the matrix pipe less efficiently) + prologue / epilogue
  const double eff = (c.bm == 256 ? 1.0 : 0.85) * (c.bn == 256 ? 1.0 : c.bn == 192 ? 0.97 : 0.88);
  const double per_tile = kt * (c.bm / 256.0) * (c.bn / 256.0) / eff + 2.5 * (c.bm / 256.0) * (c.bn / 256.0) + 1.0 + (splits > 1 ? 1.5 : 0.0);
  return rounds * per_tile;
}

// plan: configuration index and K slices
static void pp_plan(const tfpp_conv_params& p, int* cfg_out, int* splits_out) {
  const long M = (long)p.B * p.Hd * p.Wd, N = p.n_g, K = p.ks_g;
  if (g_pp_force > 0) {
    int c = (g_pp_force % 100) - 1, s = g_pp_force / 100;
    if (c < 0 || c >= kPPNumCfg) c = 0;
    if (s < 1 || !p.splitk_ws) s = 1;
    while (s > 1 && ((long)s * M * N > p.splitk_ws_floats || cdiv(K, 64) / s < 4)) --s;
    *cfg_out = c; *splits_out = s;
    return;
  }
  static const int mf32_env = [] { const char* e = std::getenv("TFPP_GEMM_PP_MF32"); return e ? std::atoi(e) : 1; }();
  double best = 1e30;
  int bc = 0, bs = 1;
  for (int c = 0; c < kPPNumCfg; ++c) {
    if (kPPCfg[c].mf32 != (mf32_env && kPPCfg[c].bn != 192 ? 1 : 0)) continue;
    for (int s = 1; s <= 4; ++s) {
      if (s > 1 && (!p.splitk_ws || (long)s * M * N > p.splitk_ws_floats || N % 4 || cdiv(K, 64) / s < 8)) break;
      const double cost = pp_cost(M, N, K, kPPCfg[c], s);
      if (cost < best) { best = cost; bc = c; bs = s; }
    }
  }
  *cfg_out = bc; *splits_out = bs;
}

int conv_pp_variant(const tfpp_conv_params& p) { int c, s; pp_plan(p, &c, &s); return 210 + c; }
int conv_pp_splits(const tfpp_conv_params& p) { int c, s; pp_plan(p, &c, &s); return s; }
int conv_pp_bm(int variant) { return kPPCfg[variant - 210].bm; }

template <int BM, int BN, bool MF32> static int launch_pp(const tfpp_conv_params& p, hipStream_t st) {
  const long M = (long)p.B * p.Hd * p.Wd;
  PPMap map;
  map.tm = cdiv(M, BM); map.tn = cdiv(p.n_g, BN); map.dbg = g_pp_dbg;
  // XCD rectangles: minimise xn * |A| + xm * |B| (bytes crossing the fabric); unbalanced rectangles and unused XCDs cost rounds
  const double a_bytes = (double)M * p.ks_g * 2, b_bytes = (double)p.n_g * p.ks_g * 2;
  double best = 1e300;
  map.xm = 1; map.xn = 1;
  for (int xm = 1; xm <= 8; xm *= 2)
    for (int xn = 1; xm * xn <= 8; xn *= 2) {
      if (xm > map.tm || xn > map.tn) continue;
      const double imb = (double)(cdiv(map.tm, xm) * cdiv(map.tn, xn)) * 8.0 / ((double)map.tm * map.tn);  // largest rectangle / mean share of an XCD
      const double cost = (xn * a_bytes + xm * b_bytes) * (0.25 + 0.75 * imb);
      if (cost < best) { best = cost; map.xm = xm; map.xn = xn; }
    }
  const int per = cdiv(map.tm, map.xm) * cdiv(map.tn, map.xn);
  dim3 grid(8 * per, p.splitk > 1 ? p.splitk : 1, 1);
  const size_t lds = (size_t)2 * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_pp_kernel<BM, BN, MF32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_gemm_pp_kernel<BM, BN, MF32>), grid, dim3(512), lds, st, p, map);
  TFPP_CHECK_LAUNCH();
  return 0;
}

int conv_gemm_pp(const tfpp_conv_params& p, hipStream_t st) {
  int c, s;
  pp_plan(p, &c, &s);
  switch (c) {
    case 0: return launch_pp<256, 256, false>(p, st);
    case 1: return launch_pp<256, 256, true>(p, st);
    case 2: return launch_pp<256, 192, false>(p, st);
    case 3: return launch_pp<256, 128, false>(p, st);
    case 4: return launch_pp<256, 128, true>(p, st);
    case 5: return launch_pp<128, 256, false>(p, st);
    case 6: return launch_pp<128, 256, true>(p, st);
    case 7: return launch_pp<128, 192, false>(p, st);
    case 8: return launch_pp<128, 128, false>(p, st);
    default: return launch_pp<128, 128, true>(p, st);
  }
}
