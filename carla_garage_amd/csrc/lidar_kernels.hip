// LiDAR point cloud -> BEV occupancy histogram (the input of the LiDAR branch), on the GPU.
// Replaces CARLA_Data.lidar_to_histogram_features (team_code/data.py:873-906; numpy histogramdd on ~60 k points + H2D copy on every
// 20 Hz tick, team_code/sensor_agent.py:421-425).  Integer work: one int32 atomic per surviving point (order-independent, hence
// bit-exact), then one pass that clips, scales and writes the transposed (C, H, W) float image.
// Bin semantics are numpy's: index = searchsorted(edges, v, side='right') - 1 on the float64 edges the reference builds with
// np.linspace (passed in, so any grid configuration bins identically), the last edge belongs to the last bin, NaN / outside dropped.
#include "common.cuh"
#include "../../include/tfpp.h"

__device__ __forceinline__ int lidar_bin(const double* __restrict__ edges, int n, double v) {
  if (!(v >= edges[0] && v <= edges[n])) return -1;  // also rejects NaN
  if (v == edges[n]) return n - 1;
  int lo = 0, hi = n;  // invariant: edges[lo] <= v < edges[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (v >= edges[mid]) lo = mid;
    else hi = mid;
  }
  return lo;
}

__global__ void lidar_scatter_kernel(const float* __restrict__ pts, long n, int stride, const double* __restrict__ xe, int nx,
                                     const double* __restrict__ ye, int ny, int* __restrict__ counts, float max_height, float split,
                                     int use_ground_plane) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = pts + i * stride;
  const float z = p[2];
  if (!(z < max_height)) return;   // data.py:895 (NaN heights fail the comparison there too)
  const bool above = z > split;    // data.py:896-897
  if (!above && !use_ground_plane) return;
  const int ix = lidar_bin(xe, nx, (double)p[0]);
  if (ix < 0) return;
  const int iy = lidar_bin(ye, ny, (double)p[1]);
  if (iy < 0) return;
  const int c = use_ground_plane ? (above ? 1 : 0) : 0;
  atomicAdd(counts + ((size_t)c * ny + iy) * nx + ix, 1);  // transposed: row = y bin, column = x bin (data.py:892)
}

__global__ void lidar_finalize_kernel(const int* __restrict__ counts, float* __restrict__ out, long total, int hist_max) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int c = counts[i];
  if (c > hist_max) c = hist_max;
  out[i] = (float)((double)c / (double)hist_max);  // float64 divide then astype(float32), as numpy does
}

extern "C" int tfpp_lidar_histogram(const float* points, int64_t n, int point_stride, const double* xedges, int nx, const double* yedges,
                                    int ny, int32_t* counts, float* out, float max_height, float split_height, int use_ground_plane,
                                    int hist_max, void* stream) {
  if ((n > 0 && !points) || !xedges || !yedges || !counts || !out || nx < 1 || ny < 1 || point_stride < 3 || hist_max < 1 || n < 0)
    return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)(use_ground_plane ? 2 : 1) * nx * ny;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)total * sizeof(int), st);
  if (e != hipSuccess) return -(int)e;
  if (n > 0)
    hipLaunchKernelGGL(lidar_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, points, (long)n, point_stride, xedges, nx,
                       yedges, ny, counts, max_height, split_height, use_ground_plane);
  hipLaunchKernelGGL(lidar_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, out, total, hist_max);
  TFPP_CHECK_LAUNCH();
  return 0;
}
