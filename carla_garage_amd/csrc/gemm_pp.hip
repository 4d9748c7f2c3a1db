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

struct PPMap { int tm, tn, xm, xn, dbg; };  // dbg: ablation bits, only in builds with -DTFPP_PP_DBG (TFPP_BUILD_FLAGS): 1 no steady-state DMA, 2 no fragment reads, 4 no MFMAs, 8 no stores

template <int OFF> __device__ __forceinline__ pp_u32x4_t pp_ds_read(unsigned addr) {  // OFF: 16-bit immediate byte offset
  pp_u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// MFMAs as inline assembly with the accumulator tied to the destination: with the builtins the register allocator renames the accumulators
// across the unrolled phases (D != C), which doubles their live ranges -- 242 VGPRs instead of 150 for the 128x256 tile, scratch spills for
// 256x256.  Operands come straight from ds_read_b128 returns (no VALU write in front), every accumulator is touched once per 8+ MFMAs.
__device__ __forceinline__ void pp_mfma16(f32x4_t& acc, const pp_u32x4_t& a, const pp_u32x4_t& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pp_mfma32(f32x16_t& acc, const pp_u32x4_t& a, const pp_u32x4_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
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

// SR: rows of an A slab per wave = rows one phase multiplies (32: 16 / 12 / 8 MFMAs 16x16x32 per phase at BN = 256 / 192 / 128; 64: twice that --
// a barrier round trip costs ~100+ cycles whatever the phase does, longer phases amortise it, at 32 more VGPRs of A fragments).
template <int BM, int BN, bool MF32, int SR = 32>
__global__ __launch_bounds__(512) void conv_gemm_pp_kernel(tfpp_conv_params p, PPMap map) {
  constexpr int WM = BM / 2, WN = BN / 4, NPH = WM / SR, PPS = SR / 32, NB = BN / 64, NA = BM / 64, LPW = NB + NA;
  constexpr int KT_BYTES = (BM + BN) * 128, A_OFF = BN * 128;
  constexpr int FN = WN / 16, FN32 = WN / 32, NKS = MF32 ? 4 : 2;
  constexpr int LATE = (NPH == 4 ? 2 : 1) * PPS;  // A loads of K tile t + 1 issued in phase 0 of tile t
  static_assert(NPH == 4 || NPH == 2, "two or four phases per K tile");
  static_assert(NA == NPH * PPS && (PPS == 1 || PPS == 2) && WN % 16 == 0 && (!MF32 || WN % 32 == 0), "tile shape");
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

  // ---- LDS-DMA bookkeeping: every piece of this lane fetches logical chunk kc of its row (the swizzle depends on the piece's parity only).
  // buffer_load ... lds through one descriptor per operand: address = base + voffset (per lane, fixed per piece) + soffset (uniform: the K tile);
  // no per-load vector arithmetic.  Chunks beyond K in the last tile take an out-of-range voffset: the load returns zeros (both operands:
  // NaN * 0 must not happen).
  const int kc = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
  const bool tz = ktail && ((kt_beg + nkt - 1) * 64 + kc * 8 >= K);  // this lane's chunk of the LAST tile lies beyond K
  unsigned b_vo[NB], a_vo[NA];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int row = (wave + 8 * k) * 8 + (lane >> 3);
    const int n = bn0 + row < p.n_g ? bn0 + row : p.n_g - 1;  // rows past the end repeat the last one: finite data, never stored
    b_vo[k] = (unsigned)n * (unsigned)K * 2u + (unsigned)kc * 16u;
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) {  // piece q = k % PPS of slab k / PPS: rows of THIS wave's group (the A image is private to a group)
    const int trow = (wave >> 2) * WM + (k / PPS) * SR + ((k % PPS) * 4 + (wave & 3)) * 8 + (lane >> 3);
    const int m = bm0 + trow < M ? bm0 + trow : M - 1;
    a_vo[k] = (unsigned)m * (unsigned)p.src_ld * 2u + (unsigned)kc * 16u;
  }
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src), 0, (unsigned)((size_t)M * p.src_ld * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (unsigned)((size_t)p.n_g * K * 2), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;  // the launcher checked that both operands are smaller than 2 GB
  const unsigned wave_u = (unsigned)wave;
  // one load: piece k of K tile tt (index within this workgroup's K range) into buffer buf; TAIL (compile time): apply the K-tail mask
  auto issue_b = [&](auto TAIL_, int k, int tt, unsigned buf) {
    constexpr bool TAIL = decltype(TAIL_)::value;
    const unsigned vo = (TAIL && tz) ? OOB : b_vo[k];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (pp_lds_void_t*)(size_t)(lds_base + buf * KT_BYTES + (wave_u + 8u * k) * 1024u), 16, (int)vo,
                                             (kt_beg + tt) * 128, 0, 0);
  };
  auto issue_a = [&](auto TAIL_, int k, int tt, unsigned buf) {
    constexpr bool TAIL = decltype(TAIL_)::value;
    const unsigned vo = (TAIL && tz) ? OOB : a_vo[k];
    // LDS piece (8 rows) of the slab-major image: slab * (2 SR / 8) + group * (SR / 8) + q * 4 + (wave & 3); its parity is that of the wave
    const unsigned piece = (unsigned)(k / PPS) * (2 * SR / 8) + (wave_u >> 2) * (SR / 8) + (unsigned)(k % PPS) * 4u + (wave_u & 3u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (pp_lds_void_t*)(size_t)(lds_base + buf * KT_BYTES + A_OFF + piece * 1024u), 16, (int)vo,
                                             (kt_beg + tt) * 128, 0, 0);
  };

  // ---- fragment read bases (per K step; the swizzle term depends on the lane only: all row offsets are multiples of 16 / 32)
  unsigned a_rd[2][NKS], b_rd[2][NKS];
  if constexpr (!MF32) {
    const int r16 = lane & 15, kg = lane >> 4, s = r16 >> 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned phys = (unsigned)((ks * 4 + kg) ^ s) * 16u;
      a_rd[0][ks] = lds_base + A_OFF + (unsigned)(wm * SR + r16) * 128u + phys;
      b_rd[0][ks] = lds_base + (unsigned)(wn * WN + r16) * 128u + phys;
      a_rd[1][ks] = a_rd[0][ks] + KT_BYTES;
      b_rd[1][ks] = b_rd[0][ks] + KT_BYTES;
    }
  } else {
    const int r32 = lane & 31, kg = lane >> 5, s = (r32 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const unsigned phys = (unsigned)((ks * 2 + kg) ^ s) * 16u;
      a_rd[0][ks] = lds_base + A_OFF + (unsigned)(wm * SR + r32) * 128u + phys;
      b_rd[0][ks] = lds_base + (unsigned)(wn * WN + r32) * 128u + phys;
      a_rd[1][ks] = a_rd[0][ks] + KT_BYTES;
      b_rd[1][ks] = b_rd[0][ks] + KT_BYTES;
    }
  }

  // ---- accumulators and fragments (16x16x32: acc[slab * 2 + i][j]; 32x32x16: acc32[slab][j])
  constexpr int FAI = MF32 ? SR / 32 : SR / 16;  // A fragments per slab and K step
  f32x4_t acc[MF32 ? 1 : WM / 16][MF32 ? 1 : FN];
  f32x16_t acc32[MF32 ? WM / 32 : 1][MF32 ? FN32 : 1];
  if constexpr (!MF32) {
#pragma unroll
    for (int i = 0; i < WM / 16; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < WM / 32; ++i)
#pragma unroll
      for (int j = 0; j < FN32; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
  }
  pp_u32x4_t fa[2][NKS][FAI], fb[NKS][MF32 ? FN32 : FN];  // fa[s]: slab s & 1 (read one phase ahead, beside the MFMAs of the slab before)
#ifdef TFPP_PP_DBG
  const int dbg = map.dbg;
#else
  constexpr int dbg = 0;
#endif
  if (dbg & 2) {  // timing experiments without fragment reads multiply zeros
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int i = 0; i < FAI; ++i) { fa[0][ks][i] = pp_u32x4_t{0, 0, 0, 0}; fa[1][ks][i] = pp_u32x4_t{0, 0, 0, 0}; }
#pragma unroll
      for (int j = 0; j < (MF32 ? FN32 : FN); ++j) fb[ks][j] = pp_u32x4_t{0, 0, 0, 0};
    }
  }
  // A fragments of slab SLAB in buffer RB -> fa[SLAB & 1]
  auto read_a = [&](auto RB_, auto SLAB_) {
    constexpr int RB = decltype(RB_)::value, SLAB = decltype(SLAB_)::value, D = SLAB & 1;
    if (dbg & 2) return;
    pp_static_for<NKS>([&](auto KS_) {
      constexpr int KS = decltype(KS_)::value;
      pp_static_for<FAI>([&](auto I_) {
        constexpr int I = decltype(I_)::value;
        fa[D][KS][I] = pp_ds_read<(SLAB * 2 * SR + I * (MF32 ? 32 : 16)) * 128>(a_rd[RB][KS]);
      });
    });
  };
  auto read_b = [&](auto RB_) {
    constexpr int RB = decltype(RB_)::value;
    if (dbg & 2) return;
    pp_static_for<NKS>([&](auto KS_) {
      constexpr int KS = decltype(KS_)::value;
      pp_static_for<(MF32 ? FN32 : FN)>([&](auto J_) {
        constexpr int J = decltype(J_)::value;
        fb[KS][J] = pp_ds_read<J * (MF32 ? 32 : 16) * 128>(b_rd[RB][KS]);
      });
    });
  };

  // One K tile: t = index within this workgroup's K range, in LDS buffer BUF.  ONE copy of the MFMA code per buffer: the end of the K range
  // (no more tiles to fetch, K-tail mask on the loads of the last tile, draining waits) is handled by wave-uniform flags and branches around the
  // loads only -- per-mode copies of the body made the register allocator copy all accumulators at every join (+90 VGPRs, scratch spills).
  // FAST (compile time): the caller guarantees t + 2 < nkt - 1 -- the steady-state copy has no scalar control flow at all.
  auto tile = [&](auto BUF_, auto FAST_, int t) {
    constexpr int BUF = decltype(BUF_)::value;
    constexpr bool FAST = decltype(FAST_)::value;
    const bool have1 = FAST || t + 1 < nkt, have2 = FAST || t + 2 < nkt;          // K tiles t + 1 / t + 2 exist
    const bool tail1 = !FAST && ktail && t + 1 == nkt - 1, tail2 = !FAST && ktail && t + 2 == nkt - 1;  // ... and are the tile with the K tail
    pp_static_for<NPH>([&](auto P_) {
      constexpr int P = decltype(P_)::value;
      // ================= L phase: B fragments of this tile; LDS-DMA of the pieces whose previous contents were read one phase ago
      if constexpr (P == 0) read_b(std::integral_constant<int, BUF>{});
      if (!(dbg & 1)) {
        if constexpr (P == 0) {  // late A slabs of K tile t + 1 (other buffer; their slots were read in the last phase(s) of tile t - 1)
          if (have1) {
            if (!tail1) {
#pragma unroll
              for (int k = NA - LATE; k < NA; ++k) issue_a(std::false_type{}, k, t + 1, BUF ^ 1);
            } else {
              asm volatile("; K tail" ::: "memory");  // (keeps the two paths apart: merged, every steady-state load would pay the select)
#pragma unroll
              for (int k = NA - LATE; k < NA; ++k) issue_a(std::true_type{}, k, t + 1, BUF ^ 1);
            }
          }
        } else if (have2) {  // K tile t + 2 into this buffer
          auto sched = [&](auto TL) {
            if constexpr (NPH == 4) {  // (PPS == 1)
              if constexpr (P == 1) {
#pragma unroll
                for (int k = 0; k < (NB < 2 ? NB : 2); ++k) issue_b(TL, k, t + 2, BUF);
              } else if constexpr (P == 2) {
#pragma unroll
                for (int k = 2; k < NB; ++k) issue_b(TL, k, t + 2, BUF);
              } else {
                issue_a(TL, 0, t + 2, BUF);
                issue_a(TL, 1, t + 2, BUF);
              }
            } else {  // two phases: all of B and slab 0 of A
#pragma unroll
              for (int k = 0; k < NB; ++k) issue_b(TL, k, t + 2, BUF);
#pragma unroll
              for (int k = 0; k < PPS; ++k) issue_a(TL, k, t + 2, BUF);
            }
          };
          if (!tail2) sched(std::false_type{});
          else {
            asm volatile("; K tail" ::: "memory");
            sched(std::true_type{});
          }
        }
      }
      // ================= L phase: everything older than one K tile of loads has landed (see the header); at the end of the K range, where
      // fewer loads are in flight, simply everything.  My fragment reads have returned.
      if (have2) pp_wait_vmcnt<LPW>();
      else pp_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ================= C phase: A fragments of the NEXT slab (the next tile's slab 0 in the last phase; past the end of the K range that is
      // a read of stale LDS bytes nobody uses), then the MFMAs of this one
      __builtin_amdgcn_s_setprio(1);
      if constexpr (P + 1 < NPH) read_a(std::integral_constant<int, BUF>{}, std::integral_constant<int, P + 1>{});
      else read_a(std::integral_constant<int, BUF ^ 1>{}, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      constexpr int CUR = P & 1;
      if (dbg & 4) {
      } else if constexpr (!MF32) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < FAI; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) pp_mfma16(acc[P * FAI + i][j], fa[CUR][ks][i], fb[ks][j]);
      } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < FAI; ++i)
#pragma unroll
            for (int j = 0; j < FN32; ++j) pp_mfma32(acc32[P * FAI + i][j], fa[CUR][ks][i], fb[ks][j]);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  if (nkt > 0) {
    using std::integral_constant;
    using T = std::true_type;
    using F = std::false_type;
    // ---- prologue: K tile 0 completely, K tile 1 without its late A slabs (phase 0 of tile 0 issues those)
    {
      if (nkt == 1) {
#pragma unroll
        for (int k = 0; k < NB; ++k) issue_b(T{}, k, 0, 0);
#pragma unroll
        for (int k = 0; k < NA; ++k) issue_a(T{}, k, 0, 0);
        pp_wait_vmcnt<0>();
      } else {
#pragma unroll
        for (int k = 0; k < NB; ++k) issue_b(F{}, k, 0, 0);
#pragma unroll
        for (int k = 0; k < NA; ++k) issue_a(F{}, k, 0, 0);
        if (nkt == 2) {
#pragma unroll
          for (int k = 0; k < NB; ++k) issue_b(T{}, k, 1, 1);
#pragma unroll
          for (int k = 0; k < NA - LATE; ++k) issue_a(T{}, k, 1, 1);
        } else {
#pragma unroll
          for (int k = 0; k < NB; ++k) issue_b(F{}, k, 1, 1);
#pragma unroll
          for (int k = 0; k < NA - LATE; ++k) issue_a(F{}, k, 1, 1);
        }
        pp_wait_vmcnt<LPW - LATE>();
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_a(integral_constant<int, 0>{}, integral_constant<int, 0>{});  // slab 0 of tile 0 (retired by the lgkmcnt(0) of the first L phase)
    }
    if (wm == 1) {  // group 1 runs one interval behind group 0
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    int t = 0;
#pragma unroll 1
    for (; t + 4 < nkt; t += 2) {  // steady state: neither tile touches the end of the K range
      tile(integral_constant<int, 0>{}, T{}, t);
      tile(integral_constant<int, 1>{}, T{}, t + 1);
    }
#pragma unroll 1
    for (; t < nkt; t += 2) {  // the last 1 .. 4 tiles
      tile(integral_constant<int, 0>{}, F{}, t);
      if (t + 1 < nkt) tile(integral_constant<int, 1>{}, F{}, t + 1);
    }
    if (wm == 0) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    // the MFMAs are opaque to the compiler's hazard recogniser: results of the last ones are read by vector instructions below
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  }

  if (dbg & 8) {  // no stores: keep the accumulators alive behind a condition no data satisfies
    float sum = 0.f;
    if constexpr (!MF32) {
#pragma unroll
      for (int i = 0; i < WM / 16; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
#pragma unroll
      for (int i = 0; i < WM / 32; ++i)
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
      for (int i = 0; i < WM / 16; ++i)
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
      for (int i = 0; i < WM / 32; ++i)
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
    for (int i = 0; i < WM / 16; ++i) {
      const int m_pass = bm0 + wm * WM + i * 16;
      epi_pass_bf16<FN, WM / 16, false>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, 0);
    }
  } else {
    constexpr int PITCH = EpiStrip<FN>::PITCH;
    const int c32 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < WM / 32; ++i)
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
// configurations: index -> (BM, BN, 32x32x16 MFMA, slab rows).  Measured on MI355X (tools/gemm_pp_micro.py, profiles/r04_gemm_pp_micro.txt):
// 256x192 wins where it fills two rounds of the chip (3840x6048x1512: 480 tiles), 128x192 (two workgroups per CU) nearly everywhere else,
// 128x128 for narrow N with a long K; 256x256, the 32x32x16 core and 64-row slabs never win on this model's shapes -- a tile costs
// ~11-13 us besides its K loop (dispatch, first-tile latency, 96-128 KB of output per CU written in one burst), which 24 K tiles do not
// amortise -- they stay selectable through tfpp_gemm_pp_config for the record.
struct PPCfg { int bm, bn, mf32, sr; };
static const PPCfg kPPCfg[] = {{256, 256, 0, 32}, {256, 256, 1, 32}, {256, 192, 0, 32}, {256, 128, 0, 32}, {128, 256, 0, 32},
                               {128, 192, 0, 32}, {128, 128, 0, 32}, {256, 192, 0, 64}};
static constexpr int kPPNumCfg = (int)(sizeof(kPPCfg) / sizeof(kPPCfg[0]));
enum { PP_256x192 = 2, PP_128x192 = 5, PP_128x128 = 6 };

// Selection (tfpp_gemm_pp_config / TFPP_GEMM_PP): 0 = kernel off (DEFAULT), -2 = automatic plan, 1 + i + 100 * splits = configuration i with
// `splits` K slices.  Off by default because the training step does not get faster with it (round 4, profiles/r04_gemm_pp_*.txt): alone
// on the chip the kernel beats the LDS-DMA ring kernels by 10-22 % on the fusion-transformer shapes, warm and cold caches alike; inside the
// replayed step (other lanes' kernels on part of the CUs, power-limited clocks) its one-or-two-round grids of 25-45 us tiles lose what
// they gained -- 3840x6048x1512 takes 128-143 us in the graph against 73 us alone and 95 us for the ring kernel's 720 shorter tiles; the step
// is 0.4-2 % slower with the automatic plan, with 128-row tiles only, forward only or data gradients only, and a dynamic tile scheduler
// (workgroups claiming tiles from per-XCD counters) made it worse still (no hardware overlap of one workgroup's prologue with another's loop).
static int g_pp_force = [] { const char* e = std::getenv("TFPP_GEMM_PP"); return e ? std::atoi(e) : 0; }();
static int g_pp_dbg = 0;  // ablation bits (timing experiments only: results are wrong with any bit set), cfg / 10000
extern "C" int tfpp_gemm_pp_config(int cfg) {
  g_pp_dbg = cfg >= 10000 ? cfg / 10000 : 0;
  g_pp_force = cfg >= 10000 ? cfg % 10000 : cfg;
  return 0;
}

bool conv_pp_supported(const tfpp_conv_params& p, int dtype) {
  if (g_pp_force == 0 || g_pp_force == -1 || dtype != TFPP_BF16) return false;
  const long M = (long)p.B * p.Hd * p.Wd;
  const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd && p.G == 1;
  if (!pointwise || p.stats_partial || p.bns_partial || p.dst_nchw || p.dst_f32) return false;
  if (((p.n_g | (int)p.dst_ld | p.ks_g | (int)p.src_ld) & 7) || ((uintptr_t)p.dst & 15) || ((uintptr_t)p.src & 15) || ((uintptr_t)p.w & 15)) return false;
  if (p.res && ((((int)p.res_ld) & 7) || ((uintptr_t)p.res & 15))) return false;
  if (M * (long)p.src_ld * 2 >= (1l << 31) || (long)p.n_g * p.ks_g * 2 >= (1l << 31)) return false;  // buffer descriptors: 32-bit offsets, 0x80000000 = out of range
  if (g_pp_force > 0) return M >= 128 && p.n_g >= 128 && p.ks_g >= 64;
  // measured against the LDS-DMA ring kernels: +10 .. +22 % on the n_embd = 1512 / 576 fusion linears and the stage-4 1x1 convolutions,
  // equal on 12288x576x576 (stage 3) -- narrow N needs a long K
  if (M < 2048 || p.ks_g < 512 || p.n_g < 512) return false;
  return p.n_g >= 1024 || p.ks_g >= 1024;
}

// time model of one launch in us (fit of the micro-benchmark: a tile costs a + b per K tile; 256-row tiles run one workgroup per CU, i.e. in
// rounds of 256, the 128-row tiles two per CU, which behaves like ~290 slots without round quantisation)
static double pp_cost(long M, long N, long K, int cfg) {
  const PPCfg& c = kPPCfg[cfg];
  const double tiles = (double)cdiv(M, c.bm) * cdiv(N, c.bn), nkt = (double)cdiv(K, 64);
  if (cfg == PP_256x192) {
    const double rounds = (double)((long)(tiles + 255) / 256);
    return rounds * (12.8 + 0.81 * nkt) * (rounds >= 2 ? 1.14 : 1.0);
  }
  const double w = tiles / 290.0;
  return (w < 1.0 ? 1.0 : w) * (10.8 + 0.596 * nkt);
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
  *splits_out = 1;  // (split-K measured slower on every shape of the model: the fp32 slices cost more than the idle CUs)
  if (N < 1024) { *cfg_out = PP_128x128; return; }
  *cfg_out = pp_cost(M, N, K, PP_256x192) < 0.95 * pp_cost(M, N, K, PP_128x192) ? PP_256x192 : PP_128x192;  // (the model is a fit: ties go to the smaller tile)
}

int conv_pp_variant(const tfpp_conv_params& p) { int c, s; pp_plan(p, &c, &s); return 210 + c; }
int conv_pp_splits(const tfpp_conv_params& p) { int c, s; pp_plan(p, &c, &s); return s; }
int conv_pp_bm(int variant) { return kPPCfg[variant - 210].bm; }

template <int BM, int BN, bool MF32, int SR = 32> static int launch_pp(const tfpp_conv_params& p, hipStream_t st) {
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
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_pp_kernel<BM, BN, MF32, SR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL((conv_gemm_pp_kernel<BM, BN, MF32, SR>), grid, dim3(512), lds, st, p, map);
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
    case 4: return launch_pp<128, 256, false>(p, st);
    case 5: return launch_pp<128, 192, false>(p, st);
    case 6: return launch_pp<128, 128, false>(p, st);
    default: return launch_pp<256, 192, false, 64>(p, st);
  }
}
