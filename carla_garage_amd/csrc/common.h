// Common device helpers for the TransFuser++ gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TFPP_WAVE 64

typedef unsigned short bf16_t;  // raw bfloat16 storage
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum { TFPP_F32 = 0, TFPP_BF16 = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_GELU = 3, ACT_TANH = 4 };

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// fp32 -> bf16, round to nearest even: the native conversion (gfx950 v_cvt_pk_bf16_f32, one instruction per PAIR; the integer
// formulation it replaces cost ~6 VALU per value and was a visible share of every latency-bound epilogue)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned f2bf_pack2(float lo, float hi) {  // (bf16(lo) | bf16(hi) << 16)
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int VEC = 4;  // elements per 16-byte vector
  static constexpr int DT = TFPP_F32;
  __device__ static __forceinline__ float to_f(float v) { return v; }
  __device__ static __forceinline__ float from_f(float v) { return v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int VEC = 8;
  static constexpr int DT = TFPP_BF16;
  __device__ static __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
  __device__ static __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

// 16-byte vector <-> float[VEC]
template <typename T> __device__ __forceinline__ void unpack16(const uint4& u, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float* f) {
  return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
  return make_uint4(f2bf_pack2(f[0], f[1]), f2bf_pack2(f[2], f[3]), f2bf_pack2(f[4], f[5]), f2bf_pack2(f[6], f[7]));
}

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* f) {
  uint4 u = *reinterpret_cast<const uint4*>(p);
  unpack16<T>(u, f);
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* f) {
  *reinterpret_cast<uint4*>(p) = pack16<T>(f);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based RNG for dropout masks (same (seed, index) -> same bit in forward and backward).
__device__ __forceinline__ unsigned hash_u32(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  x ^= x >> 31;
  return (unsigned)(x >> 32);
}
// returns scale (0 or 1/(1-p)) for element idx
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  unsigned r = hash_u32(seed * 0x100000001B3ull + idx);
  return ((float)(r >> 8) * (1.0f / 16777216.0f)) < p ? 0.f : inv_keep;
}

// Per-device one-time set-up in host launchers (dynamic-LDS attributes): a process may drive more than one GPU, so the "already
// done" flag is a bit per device ordinal, not one bool per process.
static inline bool tfpp_first_use_on_this_device(unsigned long long* mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (*mask & bit) return false;
  *mask |= bit;
  return true;
}

#define TFPP_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return -(int)e__;                  \
  } while (0)

// Byte fill as a KERNEL, never hipMemsetAsync: inside a captured hipGraph a memset becomes a memset node, and the replays of the captured
// training step were not reproducible once the weight-gradient branch carried a long backlog (round 2: the zero-fill of a gradient buffer
// followed by kernels writing into it is the first place the replays diverged, with two outcomes) -- a plain kernel node keeps stream order.
// 16-byte stores on the aligned body, byte stores on the unaligned head / tail.
static __global__ void tfpp_fill_kernel(unsigned char* __restrict__ p, unsigned v32, long head, long nvec, long tail_off, long bytes) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  const uint4 val = make_uint4(v32, v32, v32, v32);
  uint4* body = reinterpret_cast<uint4*>(p + head);
  for (long j = i; j < nvec; j += stride) body[j] = val;
  const long edge = head + (bytes - tail_off);
  for (long j = i; j < edge; j += stride) p[j < head ? j : tail_off + (j - head)] = (unsigned char)v32;
}
static inline int tfpp_fill_async(void* ptr, int value, size_t nbytes, hipStream_t st) {
  if (nbytes == 0) return 0;
  const long bytes = (long)nbytes;
  long head = (long)((16 - ((uintptr_t)ptr & 15)) & 15);
  if (head > bytes) head = bytes;
  const long nvec = (bytes - head) / 16, tail_off = head + nvec * 16;
  const unsigned b = (unsigned)value & 0xffu, v32 = b | (b << 8) | (b << 16) | (b << 24);
  long blocks = (nvec + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(tfpp_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned char*)ptr, v32, head, nvec, tail_off, bytes);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------------------
// Column-fixed thread layout for channels-last [rows, C] tensors walked in 16-byte chunks (CV = C / VEC chunks a row).
// A 256-thread workgroup is RP row-slots x SW chunk-columns with SW = CV when CV <= 256, so one pass of a workgroup
// reads RP whole rows = one contiguous span, every lane that exists is active (216/256 .. 256/256 for the RegNet
// widths), and a thread keeps its channel chunk for its whole life: per-channel parameters and accumulators sit in
// registers instead of being re-fetched per element.
// ---------------------------------------------------------------------------------------------------------------
struct ColLayout { int sw, rp, ny; };
static inline ColLayout col_layout(int CV) {
  ColLayout l;
  l.sw = CV <= 256 ? CV : 256;
  l.rp = 256 / l.sw;
  l.ny = (CV + l.sw - 1) / l.sw;
  return l;
}
// number of workgroups along rows: ~rows_per_thread rows per thread, at most max_blocks workgroups in total
static inline int col_blocks_x(long rows, const ColLayout& l, int rows_per_thread, long max_blocks, long batch = 1) {
  long nb = (rows + (long)l.rp * rows_per_thread - 1) / ((long)l.rp * rows_per_thread);
  long cap = max_blocks / ((long)l.ny * batch);
  if (cap < 1) cap = 1;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}
__device__ __forceinline__ bool col_thread(int sw, int rp, int CV, int& rr, int& cv) {
  rr = (int)threadIdx.x / sw;
  cv = (int)blockIdx.y * sw + ((int)threadIdx.x - rr * sw);
  return rr < rp && cv < CV;
}
// Sum NV per-thread accumulators over the RP row-slots of a workgroup; on return the rr == 0 threads hold the totals.
// sm: NV * 256 floats of LDS.  Inactive threads must pass zeros.
template <int NV> __device__ __forceinline__ void col_block_reduce(float (&acc)[NV], int sw, int rp, int rr, float* sm) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < NV; ++e) sm[e * 256 + tid] = acc[e];
  __syncthreads();
  for (int n = rp; n > 1;) {
    const int h = (n + 1) >> 1;
    if (rr + h < n) {
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        acc[e] += sm[e * 256 + tid + h * sw];
        sm[e * 256 + tid] = acc[e];
      }
    }
    __syncthreads();
    n = h;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Grid-wide sums in a FIXED order without a second launch ("last ticket adds up"): every workgroup publishes its partial(s), draws a
// ticket, and the workgroup that draws the last one adds all partials in index order.  WHO adds varies from run to run, the order of the
// additions does not, so the result is bit-reproducible -- unlike fp32 atomicAdd, whose order is the order of arrival.
// MI355X has one L2 per XCD and they are not coherent with each other inside a kernel: partials are written and read with agent-scope
// atomic stores / loads (write-through to / fetched from the memory side), and a wave waits for its publishes (s_waitcnt vmcnt(0)) before
// the workgroup's ticket is drawn, so a drawn ticket implies published partials.  Tickets live in caller-provided scratch that is zero
// before the first use; the last workgroup puts its ticket back to zero (tfpp_gridsum_scratch_floats, include/tfpp.h).
// ---------------------------------------------------------------------------------------------------------------
#define TFPP_GRIDSUM_TICKETS 64  /* first floats of the scratch: ticket counters; partials start behind them */
__device__ __forceinline__ void grid_publish(float* slot, float v) { __hip_atomic_store(slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float grid_fetch(const float* slot) { return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Sum of n (<= 16) published partials base[k * stride], k = 0 .. n-1, added in index order.  __hip_atomic_load compiles to one
// `global_load_dword ... sc1` followed by s_waitcnt vmcnt(0) EACH: n dependent round trips to the memory side (~1 us apiece -- a 43-partial sum
// over 3 channel passes took 130 us in the first fused squeeze-excite kernel, profiles/r05_ab_se_splitk.txt).  Here the same sc1 loads are issued
// back to back and waited for once: one round trip.
__device__ __forceinline__ float grid_fetch_sum16(const float* base, long stride, int n) {
  float v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float* q = base + (long)(k < n ? k : 0) * stride;  // (slots beyond n re-read partial 0: a valid address; their value is not added)
    asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[k]) : "v"(q) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                 "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               :
               : "memory");
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < n) t += v[k];
  return t;
}
// the same for published (float, float) pairs packed into 64-bit words: lo and hi sums of base[k * stride], k < n <= 16, in index order
__device__ __forceinline__ void grid_fetch_pair_sum16(const unsigned long long* base, long stride, int n, float& lo, float& hi) {
  unsigned long long v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const unsigned long long* q = base + (long)(k < n ? k : 0) * stride;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v[k]) : "v"(q) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                 "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               :
               : "memory");
  lo = 0.f;
  hi = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < n) {
      lo += __uint_as_float((unsigned)v[k]);
      hi += __uint_as_float((unsigned)(v[k] >> 32));
    }
}
// true (for every thread of the workgroup) in the workgroup that drew the last of `total` tickets
__device__ __forceinline__ bool grid_last_ticket(unsigned* ticket, unsigned total) {
  __shared__ unsigned s_last_ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's publishes have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last_ticket = (t == total - 1u) ? 1u : 0u;
    if (s_last_ticket) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return s_last_ticket != 0u;
}
