// Micro-test: does a row-blocked 16 B/lane store pattern (a GEMM epilogue: a few lanes per output row, rows one pitch apart) cost more
// HBM write traffic / time than a streaming store of the same bytes?  rocprofv3's WRITE_SIZE reports 2x (128 B per row and wave) and 3x
// (64 B per row and wave) the bytes of C for conv_gemm_glds_kernel while streaming kernels report exactly 1x (profiles/r02_pmc_calibration.txt).
//   mode 0: streaming, a wave writes 1024 contiguous bytes per instruction
//   mode 1: 8 lanes per row (128 B), 8 rows per instruction, rows `pitch` bytes apart; neighbouring 128-B column blocks belong to other waves
//   mode 2: 4 lanes per row (64 B), 16 rows per instruction
// build: hipcc --offload-arch=gfx950 -O3 tools/store_pattern_test.hip -o tools/bin/store_pattern_test ; run: tools/bin/store_pattern_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE> __global__ void store_pattern(uint4* __restrict__ dst, long rows, long pitch16) {  // pitch in 16-byte units
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint4 v = make_uint4(lane, (unsigned)wave, 3u, 4u);
  if (MODE == 0) {
    const long total = rows * pitch16, per_wave = 64 * 8;  // 8 instructions of 64 lanes
    const long base = wave * per_wave;
    if (base >= total) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[base + i * 64 + lane] = v;
  } else {
    constexpr int LPR = MODE == 1 ? 8 : 4, RPI = 64 / LPR;  // lanes per row, rows per instruction
    const long ctiles = pitch16 / LPR;                       // column blocks of LPR * 16 bytes
    const long tm = wave / ctiles, tn = wave - tm * ctiles;
    const long r0 = tm * 64;
    if (r0 >= rows) return;
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
      const long r = r0 + i * RPI + lane / LPR;
      if (r < rows) dst[r * pitch16 + tn * LPR + (lane % LPR)] = v;
    }
  }
}

int main() {
  const long pitch = 12288, rows = 87360;  // 1.07 GB, the C of a GEMM with N = 6144 bf16 columns
  const long bytes = rows * pitch;
  uint4* d;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 3; ++mode) {
    const long waves = mode == 0 ? bytes / (64 * 8 * 16) : (rows / 64) * (pitch / 16 / (mode == 1 ? 8 : 4));
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(store_pattern<0>, grid, block, 0, 0, d, rows, pitch / 16);
      if (mode == 1) hipLaunchKernelGGL(store_pattern<1>, grid, block, 0, 0, d, rows, pitch / 16);
      if (mode == 2) hipLaunchKernelGGL(store_pattern<2>, grid, block, 0, 0, d, rows, pitch / 16);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (it > 0 && ms < best) best = ms;
    }
    printf("mode %d: %ld bytes  %.1f us  %.0f GB/s\n", mode, bytes, best * 1e3, bytes / best * 1e-6);
  }
  return 0;
}
