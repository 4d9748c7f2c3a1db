// Micro-test: aggregate L2 -> LDS feed rate of (0) global_load_lds_dwordx4 (LDS-DMA), (1) global_load_dwordx4 + ds_write_b128 (through VGPRs),
// (2) global_load_dwordx4 only (no LDS), on an L2 / MALL-resident working set read over and over by 256 x 2 workgroups of 512 threads.
// The LDS-DMA GEMMs plateau at ~9.5 TB/s of operand feed whatever the tile shape; this isolates the load path itself.
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_feed_test.hip -o tools/bin/lds_feed_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int MODE> __global__ __launch_bounds__(512) void feed(const uint4* __restrict__ src, long n16, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;
  // each workgroup streams a 64 KB window (4096 chunks of 16 B) per iteration, windows spread over the buffer
  const long win = 4096;
  long base = ((long)blockIdx.x * 7919) % (n16 / win) * win;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint4* p = src + base + k * 512 + tid;
      if (MODE == 0) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)p, (lds_void_t*)(lds_base + (unsigned)((k * 8 + wave) * 1024)), 16, 0, 0);
      } else {
        const uint4 v = *p;
        if (MODE == 1) *reinterpret_cast<uint4*>(smem + (k * 512 + tid) * 16) = v;
        else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    base += win * 977;
    if (base >= n16 - win) base %= (n16 - win);
    base = base / win * win;
  }
  if (MODE != 2) acc = *reinterpret_cast<uint4*>(smem + tid * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const long bytes = (argc > 1 ? atol(argv[1]) : 64l) << 20;  // working set in MB (argv[1]); 64 MB: Infinity Cache resident, <= 4 MB: every L2
  uint4* d; unsigned* sink;
  hipMalloc(&d, bytes); hipMalloc(&sink, 4);
  hipMemset(d, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 200, grid = 512;
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      if (mode == 0) { hipFuncSetAttribute((const void*)feed<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(feed<0>, dim3(grid), dim3(512), 65536, 0, d, bytes / 16, iters, sink); }
      if (mode == 1) { hipFuncSetAttribute((const void*)feed<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(feed<1>, dim3(grid), dim3(512), 65536, 0, d, bytes / 16, iters, sink); }
      if (mode == 2) hipLaunchKernelGGL(feed<2>, dim3(grid), dim3(512), 65536, 0, d, bytes / 16, iters, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (rep > 0 && ms < best) best = ms;
    }
    const double moved = (double)grid * iters * 65536.0;
    printf("mode %d (%s): %.1f us  %.2f TB/s\n", mode, mode == 0 ? "global_load_lds x4" : (mode == 1 ? "global_load x4 + ds_write_b128" : "global_load x4 only"), best * 1e3,
           moved / best * 1e-9);
  }
  return 0;
}
