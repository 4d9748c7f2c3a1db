// Multi-stage LDS-DMA implicit GEMM (forward / data gradient), bf16, for the N >= 128 shapes of the model.
//
// Structure (cdna_hip_programming.md section 5, "glds" rows): every K-tile (BK = 32) of A (BM rows) and B (BN rows) is copied
// HBM/L2 -> LDS by `global_load_lds_dwordx4` (no VGPR round trip, no ds_write pass), NSTAGE tiles deep, so NSTAGE-1 tiles are
// in flight behind the MFMAs of the current one; one raw s_barrier per tile, counted `s_waitcnt vmcnt(N)` (never 0 in the
// steady state).  LDS-DMA writes lane-linear (wave-uniform base + lane*16 B), so the tile image is linear [row][4 x 16 B] and the
// bank-conflict swizzle is applied on the SOURCE chunk index and again on the fragment read (rule 21):
//     physical 16-byte slot of (row, kc) = kc ^ F[(row >> 2) & 3],  F = {0, 2, 3, 1}
// which makes every ds_read_b128 lane group hit 16 distinct slots.  LDS-DMA cannot zero-fill, so out-of-range rows / taps /
// K-tails read a 16-byte zero page in global memory.  Fragment reads are inline-asm ds_read_b128 (the compiler would otherwise
// drain the DMA queue with vmcnt(0) before any LDS read it can see).
#include "gemm_core.h"
#include "gemm_internal.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

__device__ __attribute__((aligned(16))) unsigned int tfpp_zero_page[4] = {0u, 0u, 0u, 0u};

// TFPP_GLDS_TRACE=1: phase timestamps (100 MHz wall clock) of up to 4096 workgroups, read back with tfpp_debug_glds_trace
#define TRACE_SLOTS 6
__device__ unsigned long long tfpp_glds_trace[4096 * TRACE_SLOTS];
#define TRACE(k) do { if (trace && tid == 0) { const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z); if (bid < 4096) tfpp_glds_trace[bid * TRACE_SLOTS + (k)] = wall_clock64(); } } while (0)

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 lds_read_b128_asm(unsigned addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return make_uint4(v[0], v[1], v[2], v[3]);
}

template <int OFF> __device__ __forceinline__ uint4 lds_read_b128_imm(unsigned addr) {  // OFF: 16-bit immediate byte offset
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
// N fragments STEP bytes (16 rows) apart from one base address, in issue order 0 .. N-1
template <int I, int N, int STEP> struct LdsFragReads {
  static __device__ __forceinline__ void run(Frag<bf16_t>* f, unsigned base) {
    f[I].v = lds_read_b128_imm<I * STEP>(base);
    LdsFragReads<I + 1, N, STEP>::run(f, base);
  }
};
template <int N, int STEP> struct LdsFragReads<N, N, STEP> {
  static __device__ __forceinline__ void run(Frag<bf16_t>*, unsigned) {}
};

// physical 16-byte slot of logical chunk kc in row `row` of a stage image with CPR chunks per row (CPR = 4: 64-byte rows, the
// {0,2,3,1}[(row>>2)&3] table above; CPR = 8: 128-byte rows, two rows per 256-byte bank row: XOR with (row>>1)&7 sends the 16 rows of
// a ds_read_b128 lane group to 16 distinct slots)
template <int CPR> __device__ __forceinline__ int glds_swz(int row) {
  if constexpr (CPR == 4) return (0x1320 >> (((row >> 2) & 3) * 4)) & 3;
  else return (row >> 1) & 7;
}

// KS: MFMA k-steps (of 32) per stage = per barrier.  KS = 2 halves the barriers, waits and address arithmetic per MFMA (the
// per-stage overhead of ~40 non-MFMA instructions against 8 MFMAs was the measured limit of the KS = 1 kernel).
// PW: pointwise layers (1x1, stride 1, no padding: every fusion linear and RegNet 1x1 conv -- all this kernel is used for in the model) run
// a lean K loop: 32-bit running offsets against a uniform base (row / column indices clamped into range instead of redirected to the zero
// page; only the K tail of the last tile takes the general pointer path), one vector add per load and per operand base, immediate
// offsets on the fragment reads.  The general loop below it executes ~300 instructions per tile and wave -- tools/glds_trace.py with every
// load, read and MFMA compiled out still measured 358 ns per tile, as much as the complete 8-MFMA tile should take.
template <int BM, int BN, int NSTAGE, int WGM, int WGN, int KS = 1, bool BNS = false, bool PW = false>
__global__ __launch_bounds__(WGM * WGN * 64) void conv_gemm_glds_kernel(tfpp_conv_params p, int trace, int m_major) {
  typedef bf16_t T;
  constexpr int NT = WGM * WGN * 64, NWAVES = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
  constexpr int BKT = 32 * KS, CPR = 4 * KS, ROWB = 64 * KS;  // K elements / 16-byte chunks / bytes per stage row
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_INST = BM * CPR / NT, B_INST = BN * CPR / NT;  // 16-byte chunks per thread per tile
  static_assert((BM * CPR) % NT == 0 && (BN * CPR) % NT == 0, "tile must divide over the threads");
  constexpr int LOADS = A_INST + B_INST;              // LDS-DMA instructions per wave per tile
  static_assert((FM - 1) + (FM + FN) * (KS - 1) <= 15, "lgkmcnt is a 4-bit counter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TRACE(0);
  const int wm = wave / WGN, wn = wave % WGN;
  int g = blockIdx.z, split = 0;
  if (p.splitk > 1) { g = blockIdx.z / p.splitk; split = blockIdx.z - g * p.splitk; }  // z = (group, K slice)
  const int M = p.B * p.Hd * p.Wd, K = p.R * p.S * p.ks_g;
  // Workgroup id -> XCD id % 8.  Default order (x = N-tile fastest): every XCD sees all M-tiles but only every 8th weight panel, so
  // its share of the weights stays in its L2 ("weights stationary") and the activation rows are fetched by up to 8 XCDs.  When the
  // weights are small (RegNet 1x1 convs: 0.7 MB against 14 MB of activations) that is the wrong way round: m_major gives XCD x the
  // M-tiles x, x+8, ... and walks all N-tiles of one M-tile back to back, so an activation tile enters one L2 once.
  int mtile = blockIdx.y, ntile = blockIdx.x;
  if (m_major) {
    const int gx = gridDim.x, gy = gridDim.y, id = blockIdx.x + gx * blockIdx.y;
    const int full = (gy >> 3) << 3;  // M-tiles covered by whole rounds of 8
    if (id < full * gx) {
      const int xcd = id & 7, j = id >> 3;
      ntile = j % gx;
      mtile = (j / gx) * 8 + xcd;
    } else {  // the last (< 8) M-tiles keep the default order
      const int r = id - full * gx;
      ntile = r % gx;
      mtile = full + r / gx;
    }
  }
  if constexpr (PW) {  // integer divisions run on the vector ALU: tell the compiler the results are wave-uniform (scalar loop control)
    mtile = __builtin_amdgcn_readfirstlane(mtile); ntile = __builtin_amdgcn_readfirstlane(ntile);
    g = __builtin_amdgcn_readfirstlane(g); split = __builtin_amdgcn_readfirstlane(split);
  }
  const int bm0 = mtile * BM, bn0 = ntile * BN;
  const T* __restrict__ src = reinterpret_cast<const T*>(p.src) + g * p.ks_g;
  const T* __restrict__ wk = reinterpret_cast<const T*>(p.w) + (size_t)g * p.n_g * K;
  const T* zero = reinterpret_cast<const T*>(tfpp_zero_page);
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;  // LDS byte address of the ring

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int r16 = lane & 15, kgrp = lane >> 4;

  if constexpr (PW) {
    // K tiles of this workgroup: all of them, or one slice of a split-K launch
    int kt_beg = 0, nkt = (K + BKT - 1) / BKT;
    if (p.splitk > 1) {
      const int per = (nkt + p.splitk - 1) / p.splitk;
      kt_beg = split * per;
      nkt = (kt_beg + per < nkt ? kt_beg + per : nkt) - kt_beg;
      if (nkt < 0) nkt = 0;
    }
    kt_beg = __builtin_amdgcn_readfirstlane(kt_beg);
    nkt = __builtin_amdgcn_readfirstlane(nkt);
    const int last_kt = nkt - 1;
    const bool ktail = ((kt_beg + nkt) * BKT > K);  // the last tile of this workgroup reaches past K (wave-uniform)
    const char* a_base = reinterpret_cast<const char*>(src);  // uniform
    const char* b_base = reinterpret_cast<const char*>(wk);
    unsigned a_vo[A_INST], b_vo[B_INST];  // byte offsets of this thread's chunks in the next tile to issue (the launcher checked 32 bits)
    bool a_tz[A_INST], b_tz[B_INST];      // chunk lies beyond K in the last tile
#pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int q = i * NT + tid, row = q / CPR, kc = (q % CPR) ^ glds_swz<CPR>(row);
      const int m = bm0 + row < M ? bm0 + row : M - 1;  // rows past the end repeat the last one: finite data, never stored
      a_vo[i] = (unsigned)m * (unsigned)p.src_ld * 2u + (unsigned)(kt_beg * BKT + kc * 8) * 2u;
      a_tz[i] = (kt_beg + last_kt) * BKT + kc * 8 >= K;
    }
#pragma unroll
    for (int j = 0; j < B_INST; ++j) {
      const int q = j * NT + tid, row = q / CPR, kc = (q % CPR) ^ glds_swz<CPR>(row);
      const int n = bn0 + row < p.n_g ? bn0 + row : p.n_g - 1;
      b_vo[j] = (unsigned)n * (unsigned)K * 2u + (unsigned)(kt_beg * BKT + kc * 8) * 2u;
      b_tz[j] = (kt_beg + last_kt) * BKT + kc * 8 >= K;
    }
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane(wave);
    const unsigned ring_end = lds_base + NSTAGE * STAGE_BYTES;
    auto issue = [&](unsigned stage, bool last) {  // `last` is wave-uniform; both variants issue LOADS loads per thread, in the same order
      if (!last) {
#pragma unroll
        for (int i = 0; i < A_INST; ++i) {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_base + a_vo[i]), (lds_void_t*)(stage + (i * NWAVES + wave_u) * 1024u), 16, 0, 0);
          a_vo[i] += BKT * 2;
        }
#pragma unroll
        for (int j = 0; j < B_INST; ++j) {
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_base + b_vo[j]), (lds_void_t*)(stage + A_BYTES + (j * NWAVES + wave_u) * 1024u), 16, 0, 0);
          b_vo[j] += BKT * 2;
        }
      } else {  // K % 32 != 0: chunks beyond K read the zero page (NaN * 0 must not happen on either operand)
        const char* zero = reinterpret_cast<const char*>(tfpp_zero_page);
#pragma unroll
        for (int i = 0; i < A_INST; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(a_tz[i] ? zero : a_base + a_vo[i]), (lds_void_t*)(stage + (i * NWAVES + wave_u) * 1024u), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < B_INST; ++j)
          __builtin_amdgcn_global_load_lds((gbl_void_t*)(b_tz[j] ? zero : b_base + b_vo[j]), (lds_void_t*)(stage + A_BYTES + (j * NWAVES + wave_u) * 1024u), 16, 0, 0);
      }
    };
    // fragment reads: row * ROWB + (((ks * 4 + kgrp) ^ swz(row)) * 16); the swizzle depends on row bits 1..3 only, so fragment row i /
    // column j of a wave differ from fragment 0 by i (j) * 16 rows -- an immediate
    const int arow = wm * WM + r16, brow = wn * WN + r16;
    unsigned a_off0[KS], b_off0[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      a_off0[ks] = (unsigned)(arow * ROWB + (((ks * 4 + kgrp) ^ glds_swz<CPR>(arow)) * 16));
      b_off0[ks] = (unsigned)(A_BYTES + brow * ROWB + (((ks * 4 + kgrp) ^ glds_swz<CPR>(brow)) * 16));
    }
    auto tile_mma = [&](unsigned stage) {
      Frag<T> fa[KS][FM], fb[KS][FN];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        LdsFragReads<0, FN, 16 * ROWB>::run(fb[ks], stage + b_off0[ks]);
        LdsFragReads<0, FM, 16 * ROWB>::run(fa[ks], stage + a_off0[ks]);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {  // DS operations retire in order: row i of k-step ks needs its FN weight fragments and rows 0..i
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((FM - 1 - i) + (FM + FN) * (KS - 1 - ks)) : "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; ++j) frag_mma(fa[ks][i], fb[ks][j], acc[i][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    TRACE(1);
    unsigned wr = lds_base, rd = lds_base;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nkt) { issue(wr, ktail && s == last_kt); wr += STAGE_BYTES; }
    TRACE(2);
    TRACE(3);
    int kt = 0;
#pragma unroll 1
    for (; kt + NSTAGE - 1 < nkt; ++kt) {  // steady state: NSTAGE - 2 younger tiles stay in flight
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * LOADS) : "memory");
      __builtin_amdgcn_s_barrier();  // tile kt landed for every wave; every wave is done with tile kt - 1, whose slot is refilled now
      issue(wr, ktail && kt + NSTAGE - 1 == last_kt);
      wr += STAGE_BYTES;
      if (wr == ring_end) wr = lds_base;
      tile_mma(rd);
      rd += STAGE_BYTES;
      if (rd == ring_end) rd = lds_base;
    }
#pragma unroll 1
    for (; kt < nkt; ++kt) {  // drain: nkt - 1 - kt (<= NSTAGE - 2) younger tiles in flight
      switch (nkt - 1 - kt) {  // wave-uniform
#define TFPP_WAIT_CASE(A) case A: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * LOADS < 63 ? (A) * LOADS : 63) : "memory"); break
        TFPP_WAIT_CASE(1); TFPP_WAIT_CASE(2); TFPP_WAIT_CASE(3); TFPP_WAIT_CASE(4); TFPP_WAIT_CASE(5); TFPP_WAIT_CASE(6);
#undef TFPP_WAIT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      tile_mma(rd);
      rd += STAGE_BYTES;
      if (rd == ring_end) rd = lds_base;
    }
  } else {
    // ---- per-thread chunk bookkeeping (fixed over the K loop)
    int a_kc[A_INST], a_b[A_INST], a_h0[A_INST], a_w0[A_INST];
    const T* a_ptr[A_INST];  // pointwise fast path
    const bool pointwise = (p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0);
  #pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const int q = i * NT + tid, row = q / CPR, slot = q % CPR;
      a_kc[i] = slot ^ glds_swz<CPR>(row);
      const int m = bm0 + row;
      a_ptr[i] = nullptr;
      if (m < M) {
        const int hw = p.Hd * p.Wd, b = m / hw, pix = m - b * hw, hd = pix / p.Wd, wd = pix - hd * p.Wd;
        a_b[i] = b;
        if (p.mode == 0) { a_h0[i] = hd * p.stride - p.pad; a_w0[i] = wd * p.stride - p.pad; }
        else { a_h0[i] = hd + p.pad; a_w0[i] = wd + p.pad; }
        if (pointwise) a_ptr[i] = src + ((size_t)(b * p.Hs + a_h0[i]) * p.Ws + a_w0[i]) * p.src_ld;
      } else {
        a_b[i] = -1; a_h0[i] = 0; a_w0[i] = 0;
      }
    }
    int b_kc[B_INST];
    const T* b_ptr[B_INST];
  #pragma unroll
    for (int j = 0; j < B_INST; ++j) {
      const int q = j * NT + tid, row = q / CPR, slot = q % CPR;
      b_kc[j] = slot ^ glds_swz<CPR>(row);
      const int n = bn0 + row;
      b_ptr[j] = (n < p.n_g) ? wk + (size_t)n * K : nullptr;
    }

    // K tiles of this workgroup: all of them, or one slice of a split-K launch
    int kt_beg = 0, nkt = (K + BKT - 1) / BKT;
    if (p.splitk > 1) {
      const int per = (nkt + p.splitk - 1) / p.splitk;
      kt_beg = split * per;
      nkt = (kt_beg + per < nkt ? kt_beg + per : nkt) - kt_beg;
      if (nkt < 0) nkt = 0;
    }
    // Incremental addressing (SQ_INSTS_VALU / SQ_INSTS_MFMA measured 5.8 with per-tile address generation): a chunk of a pointwise
    // layer and every weight chunk keep a running pointer that advances by one K-tile (64 bytes) per issue; rows outside the problem
    // point at the zero page with step 0, the K tail (K % 32 != 0) exists only in the last tile and is masked there.
    const int last_kt = nkt - 1;
    const bool ktail = ((kt_beg + nkt) * BKT > K);  // the last tile of this workgroup reaches past K (wave-uniform)
    const T* a_cur[A_INST];
    int a_stepe[A_INST];  // elements per tile: 32 or 0
  #pragma unroll
    for (int i = 0; i < A_INST; ++i) {
      const bool lin = pointwise && a_ptr[i] != nullptr;
      a_cur[i] = lin ? a_ptr[i] + kt_beg * BKT + a_kc[i] * 8 : zero;
      a_stepe[i] = lin ? BKT : 0;
    }
    const T* b_cur[B_INST];
    int b_stepe[B_INST];
  #pragma unroll
    for (int j = 0; j < B_INST; ++j) {
      b_cur[j] = b_ptr[j] ? b_ptr[j] + kt_beg * BKT + b_kc[j] * 8 : zero;
      b_stepe[j] = b_ptr[j] ? BKT : 0;
    }
    auto issue = [&](int kt) {  // LDS-DMA of this workgroup's K-tile kt into ring slot kt % NSTAGE; called with kt = 0, 1, 2, ... in order
      const unsigned stage = lds_base + (unsigned)((kt % NSTAGE) * STAGE_BYTES);
      const bool last = ktail && kt == last_kt;  // wave-uniform
  #pragma unroll
      for (int i = 0; i < A_INST; ++i) {
        const T* gp = a_cur[i];
        if (pointwise) {
          if (last && (kt_beg + kt) * BKT + a_kc[i] * 8 >= K) gp = zero;
          a_cur[i] += a_stepe[i];
        } else {
          const int k0 = (kt_beg + kt) * BKT + a_kc[i] * 8;
          gp = zero;
          if (k0 < K && a_b[i] >= 0) {
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
            if (ok) gp = src + ((size_t)(a_b[i] * p.Hs + hs) * p.Ws + ws) * p.src_ld + c;
          }
        }
        __builtin_amdgcn_global_load_lds((gbl_void_t*)gp, (lds_void_t*)(stage + (unsigned)((i * NWAVES + wave) * 1024)), 16, 0, 0);
      }
  #pragma unroll
      for (int j = 0; j < B_INST; ++j) {
        const T* gp = b_cur[j];
        if (last && (kt_beg + kt) * BKT + b_kc[j] * 8 >= K) gp = zero;
        b_cur[j] += b_stepe[j];
        __builtin_amdgcn_global_load_lds((gbl_void_t*)gp, (lds_void_t*)(stage + (unsigned)(A_BYTES + (j * NWAVES + wave) * 1024)), 16, 0, 0);
      }
    };

    // fragment read offsets inside a stage (fixed): row * ROWB + ((ks*4 + (l>>4)) ^ swz(row)) * 16
    unsigned a_off[KS][FM], b_off[KS][FN];
  #pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
  #pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WM + i * 16 + r16;
        a_off[ks][i] = (unsigned)(row * ROWB + (((ks * 4 + kgrp) ^ glds_swz<CPR>(row)) * 16));
      }
  #pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WN + j * 16 + r16;
        b_off[ks][j] = (unsigned)(A_BYTES + row * ROWB + (((ks * 4 + kgrp) ^ glds_swz<CPR>(row)) * 16));
      }
    }

    TRACE(1);
  #pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nkt) issue(s);
    TRACE(2);

    for (int kt = 0; kt < nkt; ++kt) {
      // tiles issued after kt and still allowed in flight: min(NSTAGE-2, nkt-1-kt)
      const int ahead = (nkt - 1 - kt) < (NSTAGE - 2) ? (nkt - 1 - kt) : (NSTAGE - 2);
      switch (ahead) {  // wave-uniform
  #define TFPP_WAIT_CASE(A) case A: asm volatile("s_waitcnt vmcnt(%0)" ::"n"((A) * LOADS < 63 ? (A) * LOADS : 63) : "memory"); break
        TFPP_WAIT_CASE(0); TFPP_WAIT_CASE(1); TFPP_WAIT_CASE(2); TFPP_WAIT_CASE(3); TFPP_WAIT_CASE(4); TFPP_WAIT_CASE(5);
        TFPP_WAIT_CASE(6); TFPP_WAIT_CASE(7); TFPP_WAIT_CASE(8); TFPP_WAIT_CASE(9); TFPP_WAIT_CASE(10);
  #undef TFPP_WAIT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // tile kt landed for every wave; every wave is done with tile kt-1
      if (kt == 0) TRACE(3);
      if (kt + NSTAGE - 1 < nkt) issue(kt + NSTAGE - 1);
      const unsigned stage = lds_base + (unsigned)((kt % NSTAGE) * STAGE_BYTES);
      Frag<T> fa[KS][FM], fb[KS][FN];
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
  #pragma unroll
        for (int i = 0; i < FM; ++i) fa[ks][i].v = lds_read_b128_asm(stage + a_off[ks][i]);
  #pragma unroll
        for (int j = 0; j < FN; ++j) fb[ks][j].v = lds_read_b128_asm(stage + b_off[ks][j]);
      }
  #pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // DS operations retire in order: the MFMAs of k-step ks wait for its own FM + FN reads only
        if (ks == KS - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((FM + FN) * (KS - 1 - ks)) : "memory");
        __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
        for (int i = 0; i < FM; ++i)
  #pragma unroll
          for (int j = 0; j < FN; ++j) frag_mma(fa[ks][i], fb[ks][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  TRACE(4);
  if (p.splitk > 1) {  // raw fp32 slice -> workspace [split][M][G*n_g]; splitk_epilogue_kernel finishes the job
    const int ntot = p.G * p.n_g;
    float* __restrict__ wsp = p.splitk_ws + (size_t)split * M * ntot;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = bn0 + wn * WN + j * 16 + (lane & 15);
          if (n < p.n_g) wsp[(size_t)m * ntot + g * p.n_g + n] = acc[i][j][r];
        }
      }
    return;
  }

  // ---- fused BatchNorm statistics (same contract as the LDS-staged kernel; the ring is free after the last barrier + MMA)
  if (p.stats_partial) {
    __syncthreads();
    float* st = reinterpret_cast<float*>(smem);  // [2][WGM][WGN][FN][16]
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
          if (m < M) { const float v = acc[i][j][r] * p.alpha; s += v; q += v * v; }
        }
      s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) {
        st[(((0 * WGM + wm) * WGN + wn) * FN + j) * 16 + lane] = s;
        st[(((1 * WGM + wm) * WGN + wn) * FN + j) * 16 + lane] = q;
      }
    }
    __syncthreads();
    if (wm == 0 && lane < 16) {
      const int ctot = p.G * p.n_g;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = bn0 + wn * WN + j * 16 + lane;
        if (n < p.n_g) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WGM; ++w2) {
            s += st[(((0 * WGM + w2) * WGN + wn) * FN + j) * 16 + lane];
            q += st[(((1 * WGM + w2) * WGN + wn) * FN + j) * 16 + lane];
          }
          float* row = p.stats_partial + (size_t)(mtile % p.stats_rows) * 2 * ctot;
          if (p.stats_store) { row[g * p.n_g + n] = s; row[ctot + g * p.n_g + n] = q; }  // one writer per cell: nothing to zero (tfpp.h)
          else { atomicAdd(row + g * p.n_g + n, s); atomicAdd(row + ctot + g * p.n_g + n, q); }
        }
      }
    }
  }

  // ---- epilogue
  if (epi_vec_ok(p)) {  // coalesced: 16-row passes through a per-wave LDS strip (the ring is free)
    __syncthreads();
    float* strip = reinterpret_cast<float*>(smem) + wave * EpiStrip<FN>::FLOATS;
    if constexpr (BNS) {  // fused BatchNorm-backward statistics (tfpp.h): its own instantiation, the plain kernel does not pay its registers
      BnsAcc<FN, FM> bns;
      bns.init(p, lane, bn0 + wn * WN, g);
#pragma unroll
      for (int i = 0; i < FM; ++i) bns.prefetch(p, lane, i, bm0 + wm * WM + i * 16, M - (bm0 + wm * WM + i * 16), bn0 + wn * WN, g);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m_pass = bm0 + wm * WM + i * 16;
        epi_pass_bf16<FN, FM, true>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, g, &bns, i);
      }
      __syncthreads();  // the strips are dead
      bns.template finish<WGM, WGN>(p, reinterpret_cast<float*>(smem), wm, wn, lane, mtile, bn0 + wn * WN, g);
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m_pass = bm0 + wm * WM + i * 16;
        epi_pass_bf16<FN, FM, false>(p, acc[i], strip, lane, m_pass, M - m_pass, bn0 + wn * WN, g);
      }
    }
    TRACE(5);
    return;
  }
  const int hw = p.Hd * p.Wd;
  const T* __restrict__ res = reinterpret_cast<const T*>(p.res);
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = bm0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = bn0 + wn * WN + j * 16 + r16;
        if (n >= p.n_g) continue;
        const int ch = g * p.n_g + n;
        float v = acc[i][j][r] * p.alpha;
        if (p.scale) v *= p.scale[ch];
        if (p.shift) v += p.shift[ch];
        if (res) v += bf2f(res[(size_t)m * p.res_ld + ch]);
        v = apply_act(v, p.act);
        size_t o;
        if (p.dst_nchw) { const int b = m / hw, pix = m - b * hw; o = ((size_t)b * p.Cd + ch) * hw + pix; }
        else o = (size_t)m * p.dst_ld + ch;
        if (p.dst_f32) reinterpret_cast<float*>(p.dst)[o] = v;
        else reinterpret_cast<T*>(p.dst)[o] = f2bf(v);
      }
    }
  }
  TRACE(5);
}

template <int BM, int BN, int NSTAGE, int WGM, int WGN, int KS = 1> static int launch_glds(const tfpp_conv_params& p, hipStream_t st) {
  const long M = (long)p.B * p.Hd * p.Wd;
  dim3 grid(cdiv(p.n_g, BN), cdiv(M, BM), p.G * (p.splitk > 1 ? p.splitk : 1));
  const size_t lds = (size_t)NSTAGE * (BM + BN) * 64 * KS;
  constexpr bool HAS_PW = true;
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    constexpr bool BNS_OK = (WGM * WGN <= 8);
    const void* fns[4] = {reinterpret_cast<const void*>(&conv_gemm_glds_kernel<BM, BN, NSTAGE, WGM, WGN, KS, false, false>),
                          reinterpret_cast<const void*>(&conv_gemm_glds_kernel<BM, BN, NSTAGE, WGM, WGN, KS, BNS_OK, false>),
                          reinterpret_cast<const void*>(&conv_gemm_glds_kernel<BM, BN, NSTAGE, WGM, WGN, KS, false, HAS_PW>),
                          reinterpret_cast<const void*>(&conv_gemm_glds_kernel<BM, BN, NSTAGE, WGM, WGN, KS, BNS_OK, HAS_PW>)};
    for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  static const int trace = [] { const char* e = std::getenv("TFPP_GLDS_TRACE"); return (e && e[0] == '1') ? 1 : 0; }();
  // small weight panels (<= 6 MB per group, measured: 1512x1512 gains, 6048x1512 loses): M-major order (see the kernel)
  static const int mm_env = [] { const char* e = std::getenv("TFPP_GLDS_M_MAJOR"); return e ? std::atoi(e) : -1; }();
  const long K = (long)p.R * p.S * p.ks_g;
  const long w_bytes = (long)p.n_g * K * 2;
  const int m_major = mm_env >= 0 ? mm_env : (w_bytes <= (6l << 20) ? 1 : 0);
  // lean loop: pointwise layer, every byte offset of one group's operands fits 32 bits (TFPP_GLDS_PW=0: the general loop, for A/B runs)
  static const int pw_env = [] { const char* e = std::getenv("TFPP_GLDS_PW"); return e ? std::atoi(e) : 1; }();
  const bool pw = HAS_PW && pw_env && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.Hs == p.Hd && p.Ws == p.Wd &&
                  M * (long)p.src_ld * 2 < (1l << 32) && w_bytes < (1l << 32);
#define TFPP_GLDS_LAUNCH(BNS_, PW_) hipLaunchKernelGGL((conv_gemm_glds_kernel<BM, BN, NSTAGE, WGM, WGN, KS, BNS_, PW_>), grid, dim3(WGM * WGN * 64), lds, st, p, trace, m_major)
  // 16-wave workgroups (128 VGPRs per lane) have no room for the statistics accumulators: conv_glds_variant does not pick them then
  constexpr bool HAS_BNS = (WGM * WGN <= 8);
  if (p.bns_partial && !HAS_BNS) return TFPP_EINVAL;
  if (pw) { if (p.bns_partial) TFPP_GLDS_LAUNCH(HAS_BNS, HAS_PW); else TFPP_GLDS_LAUNCH(false, HAS_PW); }
  else { if (p.bns_partial) TFPP_GLDS_LAUNCH(HAS_BNS, false); else TFPP_GLDS_LAUNCH(false, false); }
#undef TFPP_GLDS_LAUNCH
  TFPP_CHECK_LAUNCH();
  return 0;
}

// variant codes 200 + : 0 = 128x128, 1 = 64x128, 2 = 256x128 (1024 threads)
// All three stage 64-deep K tiles, i.e. 128-byte rows: with 32-deep tiles every LDS-DMA request fetched half a cache line and the L2
// request rate (2.4e7 requests per 3840x6048x1512 launch, ~77 % of one request per channel and clock) bounded the kernel at ~500 TFLOP/s
// whatever the MFMA / LDS schedule; whole-line requests: 128x128 651, 256x128 821 TFLOP/s (round 2, tools/gemm_micro.py).
int conv_glds_variant(const tfpp_conv_params& p) {
  const long M = (long)p.B * p.Hd * p.Wd;
  const long K = (long)p.R * p.S * p.ks_g;
  const long tiles = (long)cdiv(M, 128) * cdiv(p.n_g, 128) * p.G;
  static const int min_tiles = [] { const char* e = std::getenv("TFPP_GLDS_128_MIN_TILES"); return e ? std::atoi(e) : 256; }();
  // 256x128: a third fewer operand bytes per FLOP again; needs a long K loop (>= 16 stages) to pay for its 144 KB ring prologue and
  // >= 128 workgroups.  Measured against 128x128 (TFLOP/s): 3840x6048x1512 821 / 651, 3840x1512x6048 796 / 673, 3840x1512x1512 580 / 538,
  // 3072x1512x1512 484 / 453, 12288x576x576 535 / 530, 3840x2304x576 406 / 468.
  static const int min_k256 = [] { const char* e = std::getenv("TFPP_GLDS_256_MIN_K"); return e ? std::atoi(e) : 1024; }();
  const bool pointwise = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;
  if (pointwise && K >= min_k256 && (long)cdiv(M, 256) * cdiv(p.n_g, 128) * p.G >= 128 && !p.bns_partial) return 202;
  return tiles >= min_tiles ? 200 : 201;
}
int conv_glds_bm(int variant) { return variant == 202 ? 256 : (variant == 200 ? 128 : 64); }

bool conv_glds_supported(const tfpp_conv_params& p, int dtype) {
  static const int min_k = [] { const char* e = std::getenv("TFPP_GLDS_MIN_K"); return e ? std::atoi(e) : 200; }();  // 512 until round 2: the stage-2 1x1 convs (K = 216) gain 0.45 ms/step on the 64-deep ring
  return dtype == TFPP_BF16 && p.n_g >= 128 && p.ks_g % 8 == 0 && p.src_ld % 8 == 0 && p.R * p.S * p.ks_g >= min_k;
}

static int glds_cfg() {  // TFPP_GLDS_CFG: tuning switch (stages x wave grid x k-steps per stage)
  static const int v = [] {
    const char* e = std::getenv("TFPP_GLDS_CFG");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}

int conv_gemm_glds(const tfpp_conv_params& p, hipStream_t st) {
  const int c = glds_cfg(), var = conv_glds_variant(p);
  if (var == 202) return launch_glds<256, 128, 3, 4, 4, 2>(p, st);  // 16 waves, 3 x 48 KB
  if (var == 200) {
    switch (c) {
      case 1: return launch_glds<128, 128, 4, 2, 4, 1>(p, st);  // round-1 shape: 32-deep stages (half-line requests), 4 x 16 KB
      case 10: return launch_glds<128, 128, 3, 2, 4, 2>(p, st);  // 3 x 32 KB: one workgroup per CU
      default: return launch_glds<128, 128, 2, 2, 4, 2>(p, st);  // 8 waves, 2 x 32 KB: two workgroups per CU
    }
  }
  switch (c) {
    case 1: return launch_glds<64, 128, 3, 2, 2, 1>(p, st);  // round-1 shape
    case 31: return launch_glds<64, 128, 2, 2, 2, 2>(p, st);
    case 32: return launch_glds<64, 128, 4, 2, 2, 2>(p, st);
    default: return launch_glds<64, 128, 3, 2, 2, 2>(p, st);  // 4 waves, 3 x 24 KB: two workgroups per CU
  }
}

// debugging aid: copy the phase timestamps of the last traced launch (TFPP_GLDS_TRACE=1) to host memory
extern "C" int tfpp_debug_glds_trace(uint64_t* out, int n_blocks) {
  if (!out || n_blocks < 1 || n_blocks > 4096) return TFPP_EINVAL;
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(tfpp_glds_trace), (size_t)n_blocks * TRACE_SLOTS * sizeof(unsigned long long));
  return e == hipSuccess ? TRACE_SLOTS : -(int)e;
}
