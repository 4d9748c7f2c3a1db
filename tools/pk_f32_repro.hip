// Minimal two-kernel test of the packed-FP32 issue of round 3 (profiles/r03_replay_bisect_*): waves that compute two row
// sums with packed FP32 VALU instructions (v_pk_add_f32 / v_pk_fma_f32, op_sel cross-lane-half adds) returned a wrong sum in ~2 % of the rows
// of layernorm_bwd_kernel WHEN they shared their CU with the persistent MFMA + LDS-transpose-read weight-gradient kernel.  The library has
// been built without packed FP32 since (-fno-slp-vectorize -fno-vectorize, tests/test_kernel_resources.py greps the ISA).  This program
// isolates the pair:
//   A_packed : one wave per row of 1536 fp32 pairs: s0 = sum a, s1 = sum a*b with v_pk_add_f32 / v_pk_fma_f32 + the op_sel horizontal add
//   A_scalar : the same sums in the same order with v_add_f32 / v_fma_f32 (bit-identical by construction)
//   B        : one persistent workgroup per CU looping over ds_read_b64_tr_b16 + v_mfma_f32_16x16x32_bf16 (what the co-runner was made of)
// and counts rows whose packed result differs from the scalar one: A alone, A beside B, scalar A beside B (control).
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/pk_f32_repro tools/pk_f32_repro.hip && /tmp/pk_f32_repro
// (Only explicit inline asm decides which instructions run: the compiler's own vectoriser is irrelevant here.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));

constexpr int PAIRS_PER_LANE = 12;            // 64 lanes x 12 pairs x 2 = 1536 elements per row
constexpr int ROW = 64 * PAIRS_PER_LANE * 2;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int rows) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f2* pa = reinterpret_cast<const f2*>(a + (size_t)row * ROW);
  const f2* pb = reinterpret_cast<const f2*>(b + (size_t)row * ROW);
  f2 av[PAIRS_PER_LANE], bv[PAIRS_PER_LANE];
#pragma unroll
  for (int k = 0; k < PAIRS_PER_LANE; ++k) { av[k] = pa[k * 64 + lane]; bv[k] = pb[k * 64 + lane]; }
  float s0, s1;
  if (PACKED) {
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PAIRS_PER_LANE; ++k) {
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc0) : "v"(av[k]));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(av[k]), "v"(bv[k]));
    }
    // horizontal add of the two halves with op_sel (x + y in both halves): the form the SLP vectoriser produced in layernorm_bwd_kernel
    f2 h0, h1;
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(h0) : "v"(acc0));
    asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(h1) : "v"(acc1));
    s0 = h0.x;
    s1 = h1.x;
  } else {
    float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f;
#pragma unroll
    for (int k = 0; k < PAIRS_PER_LANE; ++k) {
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(av[k].x));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(y0) : "v"(av[k].y));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(av[k].x), "v"(bv[k].x));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(y1) : "v"(av[k].y), "v"(bv[k].y));
    }
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x0), "v"(y0));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x1), "v"(y1));
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) { out[2 * row] = s0; out[2 * row + 1] = s1; }
}

// persistent co-runner: per wave, a loop of LDS transpose reads feeding MFMAs (64 KB of LDS per workgroup so that exactly one workgroup of B
// and several of A share a CU)
__global__ __launch_bounds__(256) void mfma_tr_kernel(float* __restrict__ sink, int iters) {
  extern __shared__ unsigned short lds[];
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (unsigned short)(0x3f80 + (i & 7));  // bf16 values around 1.0
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(wave * 16384 + (lane & 15) * 64 + (lane >> 4) * 8) & 0xfff8u;
  for (int it = 0; it < iters; ++it) {
    const unsigned addr = (base + (unsigned)(it & 7) * 1024u) & 0xfff8u;
    typedef int i2 __attribute__((ext_vector_type(2)));
    i2 r0, r1, r2, r3;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %4 offset:1024\n\tds_read_b64_tr_b16 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr) : "memory");
    s8 fa, fb;
    int tmp[4] = {r0.x, r0.y, r1.x, r1.y};
    int tmq[4] = {r2.x, r2.y, r3.x, r3.y};
    memcpy(&fa, tmp, 16);
    memcpy(&fb, tmq, 16);
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fa), __builtin_bit_cast(bf8, fb), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, fb), __builtin_bit_cast(bf8, fa), acc1, 0, 0, 0);
  }
  if (acc0.x + acc1.y == 12345.678f) sink[blockIdx.x] = acc0.x;  // keep the loop alive
}

static int mismatches(const std::vector<float>& got, const std::vector<float>& ref, int rows, int* first) {
  int n = 0;
  *first = -1;
  for (int r = 0; r < rows; ++r)
    if (memcmp(&got[2 * r], &ref[2 * r], 8) != 0) { if (*first < 0) *first = r; ++n; }
  return n;
}

int main(int argc, char** argv) {
  const int rows = 15360, launches = argc > 1 ? atoi(argv[1]) : 40, iters = argc > 2 ? atoi(argv[2]) : 60000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s (%s), %d CUs; %d rows of %d fp32, %d launches per case, co-runner %d iterations per workgroup\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, rows, ROW, launches, iters);
  std::vector<float> ha((size_t)rows * ROW), hb((size_t)rows * ROW);
  unsigned s = 12345u;
  for (size_t i = 0; i < ha.size(); ++i) {
    s = s * 1664525u + 1013904223u; ha[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
    s = s * 1664525u + 1013904223u; hb[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
  }
  float *a, *b, *out, *sink;
  CHECK(hipMalloc(&a, ha.size() * 4)); CHECK(hipMalloc(&b, hb.size() * 4)); CHECK(hipMalloc(&out, (size_t)rows * 8)); CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa)); CHECK(hipStreamCreate(&sb));
  CHECK(hipFuncSetAttribute((const void*)mfma_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<float> ref((size_t)rows * 2), got((size_t)rows * 2);
  hipLaunchKernelGGL(rowsum_kernel<false>, dim3(rows / 4), dim3(256), 0, sa, a, b, out, rows);
  CHECK(hipStreamSynchronize(sa));
  CHECK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
  double host0 = 0.0;  // sanity of the reference itself: row 0 against a double sum
  for (int i = 0; i < ROW; ++i) host0 += ha[i];
  printf("reference row 0: s0 = %.7e (host double sum %.7e)\n", ref[0], host0);

  struct Case { const char* name; bool packed, corun; } cases[] = {{"A_packed alone", true, false}, {"A_scalar beside B (control)", false, true},
                                                                    {"A_packed beside B", true, true}, {"A_packed alone, again", true, false}};
  int total_packed_corun = 0;
  for (const Case& c : cases) {
    int bad_launches = 0, bad_rows = 0, first_row = -1;
    for (int l = 0; l < launches; ++l) {
      CHECK(hipMemsetAsync(out, 0xff, (size_t)rows * 8, sa));
      if (c.corun) hipLaunchKernelGGL(mfma_tr_kernel, dim3(prop.multiProcessorCount), dim3(256), 65536, sb, sink, iters);
      if (c.packed) hipLaunchKernelGGL(rowsum_kernel<true>, dim3(rows / 4), dim3(256), 0, sa, a, b, out, rows);
      else hipLaunchKernelGGL(rowsum_kernel<false>, dim3(rows / 4), dim3(256), 0, sa, a, b, out, rows);
      CHECK(hipStreamSynchronize(sa));
      CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
      CHECK(hipStreamSynchronize(sb));
      int first;
      const int n = mismatches(got, ref, rows, &first);
      if (n) { ++bad_launches; bad_rows += n; if (first_row < 0) first_row = first; }
    }
    printf("%-32s: %d of %d launches with differing rows, %d rows in total (%.3f %% of all rows)%s\n", c.name, bad_launches, launches, bad_rows,
           100.0 * bad_rows / ((double)rows * launches), first_row >= 0 ? "" : "  -- bit-identical to the scalar reference");
    if (c.packed && c.corun) total_packed_corun = bad_rows;
  }
  printf("RESULT packed-FP32 corruption beside the MFMA / LDS-transpose co-runner: %s\n",
         total_packed_corun ? "REPRODUCED by this minimal pair" : "NOT reproduced by this minimal pair (the library stays built without packed FP32: the original kernel pair did show it)");
  return 0;
}
