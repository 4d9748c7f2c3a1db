// LiDAR point cloud -> BEV occupancy histogram (the input of the LiDAR branch), on the GPU.
// Replaces CARLA_Data.lidar_to_histogram_features (team_code/data.py:873-906; numpy histogramdd on ~60 k points + H2D copy on every
// 20 Hz tick, team_code/sensor_agent.py:421-425).  Integer work: one int32 atomic per surviving point (order-independent, hence
// bit-exact), then one pass that clips, scales and writes the transposed (C, H, W) float image.
// Bin semantics are numpy's: index = searchsorted(edges, v, side='right') - 1 on the float64 edges the reference builds with
// np.linspace (passed in, so any grid configuration bins identically), the last edge belongs to the last bin, NaN / outside dropped.
#include "common.h"
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
  const int e = tfpp_fill_async(counts, 0, (size_t)total * sizeof(int), st);
  if (e != 0) return e;
  if (n > 0)
    hipLaunchKernelGGL(lidar_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, points, (long)n, point_stride, xedges, nx,
                       yedges, ny, counts, max_height, split_height, use_ground_plane);
  hipLaunchKernelGGL(lidar_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, out, total, hist_max);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The loader's LiDAR path on the device: CARLA_Data.align (team_code/data.py:840-871: two
// transfuser_utils.algin_lidar calls, transfuser_utils.py:116-130 -- ego motion, then the augmentation shift / rotation) followed by
// lidar_to_histogram_features (data.py:873-906) for EVERY frame of a batch in one launch chain.  The loader workers spend ~4 ms per
// frame in numpy on it; here the raw float64 sweeps (laspy .xyz, data.py:365) are uploaded as they are.
// Arithmetic is the reference's, in float64: (p - t) elementwise, then the 3 x 3 rotation as numpy's matmul evaluates it -- an FMA chain
// over k in ascending order: acc = r0 * d0 (rounded), acc = fma(r1, d1, acc), acc = fma(r2, d2, acc) (checked against numpy bit for bit,
// oracle/lidar_port.align) -- with cos / sin taken on the host (numpy), so no transcendental is evaluated here.  xf: 10 doubles per frame
// = (t1x, t1y, t1z, cos1, sin1, t2x, t2y, t2z, cos2, sin2).  Points are concatenated; offsets[f] .. offsets[f + 1] belong to frame f.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lidar_algin(double& x, double& y, double& z, const double* __restrict__ t, double c, double s) {
#pragma clang fp contract(off)
  const double d0 = x - t[0], d1 = y - t[1], d2 = z - t[2];
  // rows of rotation_matrix.T = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
  x = __builtin_fma(0.0, d2, __builtin_fma(s, d1, c * d0));
  y = __builtin_fma(0.0, d2, __builtin_fma(c, d1, (-s) * d0));
  z = __builtin_fma(1.0, d2, __builtin_fma(0.0, d1, 0.0 * d0));
}

__global__ void lidar_align_scatter_kernel(const double* __restrict__ pts, const long long* __restrict__ offsets, const double* __restrict__ xf,
                                           int frames, const double* __restrict__ xe, int nx, const double* __restrict__ ye, int ny,
                                           int* __restrict__ counts, double max_height, double split, int use_ground_plane,
                                           double* __restrict__ aligned_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= offsets[frames]) return;
  int lo = 0, hi = frames;  // offsets[lo] <= i < offsets[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (i >= offsets[mid]) lo = mid;
    else hi = mid;
  }
  const int f = lo;
  double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  const double* q = xf + 10 * f;
  lidar_algin(x, y, z, q, q[3], q[4]);       // into the coordinate system of the current frame (data.py:863)
  lidar_algin(x, y, z, q + 5, q[8], q[9]);   // augmentation shift / rotation (data.py:865-868)
  if (aligned_out) {
    aligned_out[3 * i] = x; aligned_out[3 * i + 1] = y; aligned_out[3 * i + 2] = z;
  }
  if (!(z < max_height)) return;   // data.py:895, in float64 here (the aligned cloud is float64)
  const bool above = z > split;
  if (!above && !use_ground_plane) return;
  const int ix = lidar_bin(xe, nx, x);
  if (ix < 0) return;
  const int iy = lidar_bin(ye, ny, y);
  if (iy < 0) return;
  const int C = use_ground_plane ? 2 : 1, c = use_ground_plane ? (above ? 1 : 0) : 0;
  atomicAdd(counts + (((size_t)f * C + c) * ny + iy) * nx + ix, 1);
}

extern "C" int tfpp_lidar_align_histogram(const double* points, const int64_t* offsets, int64_t total_points, const double* xforms, int frames,
                                          const double* xedges, int nx, const double* yedges, int ny, int32_t* counts, float* out,
                                          double max_height, double split_height, int use_ground_plane, int hist_max, double* aligned_out,
                                          void* stream) {
  if ((total_points > 0 && !points) || !offsets || !xforms || frames < 1 || !xedges || !yedges || !counts || !out || nx < 1 || ny < 1 || hist_max < 1 ||
      total_points < 0)
    return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)frames * (use_ground_plane ? 2 : 1) * nx * ny;
  const int e = tfpp_fill_async(counts, 0, (size_t)total * sizeof(int), st);
  if (e != 0) return e;
  if (total_points > 0)
    hipLaunchKernelGGL(lidar_align_scatter_kernel, dim3((unsigned)((total_points + 255) / 256)), dim3(256), 0, st, points, (const long long*)offsets, xforms,
                       frames, xedges, nx, yedges, ny, counts, max_height, split_height, use_ground_plane, aligned_out);
  hipLaunchKernelGGL(lidar_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, counts, out, total, hist_max);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// CenterNet heat-map decode: LidarCenterNetHead.decode_heatmap (team_code/center_net.py:172-237) with
// get_local_maximum / get_topk_from_heatmap / transpose_and_gather_feat (team_code/gaussian_target.py:186-264) in one launch.
// One 1024-thread workgroup per sample: every thread keeps its share of the ncls*H*W candidates (3x3 local maxima keep their score,
// everything else scores 0) in registers as 64-bit keys (order-preserving score bits << 32 | ~index, so equal scores resolve to the
// lower index); k rounds of thread-max -> wave shuffle max -> LDS max pick the top-k in descending order, and the thread that owns a
// pick gathers its box (argmax over the yaw bins, class2angle, offsets, scaling) and writes row i.  fp32 operations are the
// reference's, one rounding each (no FMA contraction), so the result is bit-exact whenever the scores are distinct.
// ---------------------------------------------------------------------------------------------------------------
#define DEC_THREADS 1024
#define DEC_MAX_PER_THREAD 32
__device__ __forceinline__ unsigned long long dec_key(float v, unsigned idx) {
  unsigned b = __float_as_uint(v);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // order-preserving map of IEEE floats onto unsigned
  return ((unsigned long long)b << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ unsigned long long dec_max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

__global__ __launch_bounds__(DEC_THREADS) void centernet_decode_kernel(const float* __restrict__ heat, const float* __restrict__ wh,
                                                                       const float* __restrict__ offset, const float* __restrict__ yaw_class,
                                                                       const float* __restrict__ yaw_res, float* __restrict__ out, int ncls,
                                                                       int H, int W, int k, int nbins, float width_ratio, float height_ratio) {
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hw = H * W, n = ncls * hw;
  const float* hb = heat + (size_t)b * n;
  unsigned long long keys[DEC_MAX_PER_THREAD];
#pragma unroll
  for (int j = 0; j < DEC_MAX_PER_THREAD; ++j) {
    const int idx = tid + j * DEC_THREADS;
    keys[j] = 0ull;  // below every real candidate
    if (idx < n) {
      const int c = idx / hw, p = idx - c * hw, y = p / W, x = p - y * W;
      const float v = hb[idx];
      float m = v;  // 3x3 max pooling, padding never wins (gaussian_target.py:197-198)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int yy = y + dy, xx = x + dx;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, hb[c * hw + yy * W + xx]);
        }
      keys[j] = dec_key(m == v ? v : v * 0.f, (unsigned)idx);  // heat * keep
    }
  }
  __shared__ unsigned long long sm[DEC_THREADS / 64];
  __shared__ unsigned long long winner;
  for (int i = 0; i < k; ++i) {
    unsigned long long best = 0ull;
#pragma unroll
    for (int j = 0; j < DEC_MAX_PER_THREAD; ++j) best = dec_max(best, keys[j]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = dec_max(best, (unsigned long long)__shfl_xor((long long)best, o, 64));
    if (lane == 0) sm[wave] = best;
    __syncthreads();
    if (tid == 0) {
      unsigned long long w = sm[0];
      for (int q = 1; q < DEC_THREADS / 64; ++q) w = dec_max(w, sm[q]);
      winner = w;
    }
    __syncthreads();
    const unsigned long long w = winner;
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(w & 0xFFFFFFFFull);
    if (w != 0ull && (int)(idx % DEC_THREADS) == tid) {  // this thread owns the pick: retire it and write the box
#pragma unroll
      for (int j = 0; j < DEC_MAX_PER_THREAD; ++j)
        if (keys[j] == w) keys[j] = 0ull;
      unsigned sb = (unsigned)(w >> 32);
      sb = (sb & 0x80000000u) ? (sb & 0x7FFFFFFFu) : ~sb;
      const float score = __uint_as_float(sb);
      const int cls = (int)idx / hw, p = (int)idx - cls * hw, y = p / W, x = p - y * W;
      auto at = [&](const float* f, int C, int c) { return f[((size_t)b * C + c) * hw + p]; };
      int ycls = 0;
      float ymax = at(yaw_class, nbins, 0);
      for (int c = 1; c < nbins; ++c) { const float v = at(yaw_class, nbins, c); if (v > ymax) { ymax = v; ycls = c; } }  // first maximum
      float center = (float)ycls * (float)(2.0 * 3.141592653589793 / (double)nbins);  // class2angle, center_net.py:135-137
      asm volatile("" : "+v"(center));  // keep the product rounded on its own: the reference adds the residual in a second operation
      float angle = center + at(yaw_res, 1, 0);
      if (angle > (float)3.141592653589793) angle = angle - (float)(2.0 * 3.141592653589793);
      float* o = out + ((size_t)b * k + i) * 9;
      o[0] = ((float)x + at(offset, 2, 0)) * width_ratio;  // add then multiply: two roundings with or without contraction
      o[1] = ((float)y + at(offset, 2, 1)) * height_ratio;
      o[2] = at(wh, 2, 0) * width_ratio;
      o[3] = at(wh, 2, 1) * height_ratio;
      o[4] = angle;
      o[5] = 0.f;  // velocity / brake heads only exist for multi-frame inputs (center_net.py:214-220)
      o[6] = 0.f;
      o[7] = (float)cls;
      o[8] = score;
    }
    __syncthreads();
  }
}

extern "C" int tfpp_centernet_decode(const float* heat, const float* wh, const float* offset, const float* yaw_class, const float* yaw_res,
                                     float* out, int B, int ncls, int H, int W, int k, int num_dir_bins, float width_ratio,
                                     float height_ratio, void* stream) {
  if (!heat || !wh || !offset || !yaw_class || !yaw_res || !out || B < 1 || ncls < 1 || H < 1 || W < 1 || k < 1 || num_dir_bins < 1)
    return TFPP_EINVAL;
  const long n = (long)ncls * H * W;
  if (n > (long)DEC_THREADS * DEC_MAX_PER_THREAD || k > n) return TFPP_EINVAL;
  hipLaunchKernelGGL(centernet_decode_kernel, dim3((unsigned)B), dim3(DEC_THREADS), 0, (hipStream_t)stream, heat, wh, offset, yaw_class, yaw_res,
                     out, ncls, H, W, k, num_dir_bins, width_ratio, height_ratio);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// CenterNet training targets from the box list, on the GPU (the target rasterisation of the loader
// workers).  Replaces CARLA_Data.get_targets (team_code/data.py:697-790) with gaussian_radius / gen_gaussian_target / gaussian2d
// (team_code/gaussian_target.py:11-61,166-187) and angle2class (team_code/center_net.py:240-254).  The reference runs it on the
// unpadded float64 rows of parse_bounding_boxes, so the scalar path is float64 here too with FMA contraction off (bit-exact maps);
// only the gaussian patch is float32 (expf: within 2 ulp of numpy's).  One workgroup per sample, boxes in list order (a later box
// overwrites the scalar targets of an earlier one in the same cell, the heat-map keeps the maximum).
constexpr int TG_THREADS = 256;

__device__ double tg_gaussian_radius(double height, double width, double mo) {
#pragma clang fp contract(off)
  const double b1 = height + width;
  const double c1 = width * height * (1.0 - mo) / (1.0 + mo);
  const double r1 = (b1 - sqrt(b1 * b1 - 4.0 * c1)) / 2.0;
  const double b2 = 2.0 * (height + width);
  const double c2 = (1.0 - mo) * width * height;
  const double r2 = (b2 - sqrt(b2 * b2 - 16.0 * c2)) / 8.0;
  const double a3 = 4.0 * mo;
  const double b3 = -2.0 * mo * (height + width);
  const double c3 = (mo - 1.0) * width * height;
  const double r3 = (b3 + sqrt(b3 * b3 - 4.0 * a3 * c3)) / (2.0 * a3);
  return fmin(r1, fmin(r2, r3));
}

// numpy's float64 remainder / floor_divide (npy_divmod): the result takes the sign of the divisor
__device__ double tg_mod(double a, double b) {
#pragma clang fp contract(off)
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0.0) != (m < 0.0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}
__device__ double tg_floordiv(double a, double b) {
#pragma clang fp contract(off)
  double m = fmod(a, b);
  double div = (a - m) / b;
  if (m != 0.0 && ((b < 0.0) != (m < 0.0))) div -= 1.0;
  if (div == 0.0) return copysign(0.0, a / b);
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

__global__ __launch_bounds__(TG_THREADS) void centernet_targets_kernel(const double* __restrict__ boxes, const int* __restrict__ counts,
                                                                       float* __restrict__ heat, float* __restrict__ wh,
                                                                       float* __restrict__ offset, long long* __restrict__ yaw_class,
                                                                       float* __restrict__ yaw_res, float* __restrict__ velocity,
                                                                       long long* __restrict__ brake, float* __restrict__ pw,
                                                                       float* __restrict__ avg, int max_boxes, int ncls, int H, int W,
                                                                       int nbins, double wr, double hr, double mo) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, tid = threadIdx.x, hw = H * W;
  heat += (size_t)b * ncls * hw;
  wh += (size_t)b * 2 * hw;
  offset += (size_t)b * 2 * hw;
  pw += (size_t)b * 2 * hw;
  yaw_class += (size_t)b * hw;
  yaw_res += (size_t)b * hw;
  velocity += (size_t)b * hw;
  brake += (size_t)b * hw;
  for (int i = tid; i < ncls * hw; i += TG_THREADS) heat[i] = 0.f;
  for (int i = tid; i < 2 * hw; i += TG_THREADS) wh[i] = offset[i] = pw[i] = 0.f;
  for (int i = tid; i < hw; i += TG_THREADS) {
    yaw_class[i] = 0;
    brake[i] = 0;
    yaw_res[i] = velocity[i] = 0.f;
  }
  __syncthreads();
  const int n = min(counts[b], max_boxes);
  const double two_pi = 2.0 * 3.141592653589793;
  for (int j = 0; j < n; ++j) {
    const double* bx = boxes + ((size_t)b * max_boxes + j) * 8;
    const double ctx = bx[0] * wr, cty = bx[1] * hr;
    const int x = (int)ctx, y = (int)cty, cls = (int)bx[7];  // astype(int): toward zero
    if (x < 0 || x >= W || y < 0 || y >= H || cls < 0 || cls >= ncls) continue;  // the reference would raise / wrap; block-uniform branch
    const double ex = bx[2] * wr, ey = bx[3] * hr;
    const double rr = tg_gaussian_radius(ey, ex, mo);
    const int radius = rr >= 3.0 ? (rr < 4096.0 ? (int)rr : 4096) : 2;  // max(2, int(radius))
    const double sigma = (double)(2 * radius + 1) / 6.0;
    const float den = (float)(2.0 * sigma * sigma);
    const int left = min(x, radius), right = min(W - x, radius + 1), top = min(y, radius), bottom = min(H - y, radius + 1);
    const int rw = left + right, rh = top + bottom;
    float* hc = heat + (size_t)cls * hw;
    for (int i = tid; i < rw * rh; i += TG_THREADS) {
      const int yy = i / rw, dy = yy - top, dx = i - yy * rw - left;
      float g = expf(-(float)(dx * dx + dy * dy) / den);
      if (g < 1.1920929e-7f) g = 0.f;  // gaussian_target.py:28 (the patch maximum is exp(0) = 1)
      float* p = hc + (y + dy) * W + (x + dx);
      *p = fmaxf(*p, g);
    }
    if (tid == 0) {
      const int c = y * W + x;
      wh[c] = (float)ex;
      wh[hw + c] = (float)ey;
      const double per = two_pi / (double)nbins;
      const double shifted = tg_mod(tg_mod(bx[4], two_pi) + per / 2.0, two_pi);
      const double k = tg_floordiv(shifted, per);
      yaw_class[c] = (long long)k;
      yaw_res[c] = (float)(shifted - (k * per + per / 2.0));
      velocity[c] = (float)bx[5];
      brake[c] = (long long)rint(bx[6]);  // int(round(.)): half to even
      offset[c] = (float)(ctx - (double)x);
      offset[hw + c] = (float)(cty - (double)y);
      pw[c] = pw[hw + c] = 1.f;
    }
    __syncthreads();
  }
  __shared__ int ones;
  if (tid == 0) ones = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < ncls * hw; i += TG_THREADS) mine += heat[i] == 1.f;
  if (mine) atomicAdd(&ones, mine);
  __syncthreads();
  if (tid == 0) avg[b] = (float)max(1, ones);
}

extern "C" int tfpp_centernet_targets(const double* boxes, const int32_t* counts, float* heat, float* wh, float* offset, int64_t* yaw_class,
                                      float* yaw_res, float* velocity, int64_t* brake, float* pixel_weight, float* avg_factor, int B,
                                      int max_boxes, int ncls, int H, int W, int num_dir_bins, double width_ratio, double height_ratio,
                                      double min_overlap, void* stream) {
  if (!boxes || !counts || !heat || !wh || !offset || !yaw_class || !yaw_res || !velocity || !brake || !pixel_weight || !avg_factor || B < 1 ||
      max_boxes < 1 || ncls < 1 || H < 1 || W < 1 || num_dir_bins < 1 || !(min_overlap > 0.0 && min_overlap < 1.0))
    return TFPP_EINVAL;
  hipLaunchKernelGGL(centernet_targets_kernel, dim3((unsigned)B), dim3(TG_THREADS), 0, (hipStream_t)stream, boxes, counts, heat, wh, offset,
                     (long long*)yaw_class, yaw_res, velocity, (long long*)brake, pixel_weight, avg_factor, max_boxes, ncls, H, W, num_dir_bins,
                     width_ratio, height_ratio, min_overlap);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Rotated-box IoU + greedy non-maximum suppression on the device (replaces the shapely polygons and the
// numpy loop of team_code/transfuser_utils.py:409-450, called from sensor_agent.py:491 on the boxes of every model of the ensemble).
// One workgroup: corners in float64 (rect_polygon: half extents, rotation about the centre, translation), boxes ranked by confidence
// (descending; equal confidences: higher index first = the order in which the reference pops a stably sorted list), then the greedy loop:
// the most confident box still alive is kept and every thread clips it against one or two of the remaining boxes (Sutherland-Hodgman in
// float64, work polygons in LDS) and clears those whose IoU exceeds the threshold.  keep: indices of the kept boxes, most confident first.
// ---------------------------------------------------------------------------------------------------------------
#define NMS_THREADS 256
#define NMS_MAX_BOXES 1024

namespace {
struct NmsPoly { double x[8], y[8]; };

__device__ __forceinline__ double nms_area2(const double* x, const double* y, int n) {
#pragma clang fp contract(off)
  double s = 0.0;
  for (int i = 0; i < n; ++i) {
    const int j = (i + 1 == n) ? 0 : i + 1;
    s += x[i] * y[j] - y[i] * x[j];
  }
  return s;
}

// corners of box a in (ax, ay), of box b in (bx, by) (any orientation); w0 / w1: two work polygons of this thread in LDS
__device__ double nms_iou(const double* ax, const double* ay, const double* bx, const double* by, NmsPoly* w0, NmsPoly* w1) {
#pragma clang fp contract(off)
  const double sa = nms_area2(ax, ay, 4), sb = nms_area2(bx, by, 4);
  int n = 4;
  for (int i = 0; i < 4; ++i) { w0->x[i] = ax[i]; w0->y[i] = ay[i]; }
  NmsPoly* in = w0;
  NmsPoly* out = w1;
  for (int e = 0; e < 4 && n > 0; ++e) {
    // clip edge e of b, walked counter-clockwise (a negative half extent flips the ring of rect_polygon)
    const int i0 = sb >= 0.0 ? e : 3 - e, i1 = sb >= 0.0 ? (e + 1) & 3 : (3 - e + 3) & 3;
    const double px = bx[i0], py = by[i0], ex = bx[i1] - px, ey = by[i1] - py;
    int m = 0;
    for (int j = 0; j < n; ++j) {
      const int jp = j == 0 ? n - 1 : j - 1;
      const double cx = in->x[j], cy = in->y[j], qx = in->x[jp], qy = in->y[jp];
      const double sc = ex * (cy - py) - ey * (cx - px), sp = ex * (qy - py) - ey * (qx - px);
      if (sc >= 0.0) {
        if (sp < 0.0) {
          const double t = sp / (sp - sc);
          out->x[m] = qx + t * (cx - qx); out->y[m] = qy + t * (cy - qy); ++m;
        }
        out->x[m] = cx; out->y[m] = cy; ++m;
      } else if (sp >= 0.0) {
        const double t = sp / (sp - sc);
        out->x[m] = qx + t * (cx - qx); out->y[m] = qy + t * (cy - qy); ++m;
      }
    }
    n = m;
    NmsPoly* tmp = in; in = out; out = tmp;
  }
  const double inter = n >= 3 ? 0.5 * fabs(nms_area2(in->x, in->y, n)) : 0.0;
  const double uni = 0.5 * fabs(sa) + 0.5 * fabs(sb) - inter;
  return uni > 0.0 ? inter / uni : 0.0;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_rotated_kernel(const float* __restrict__ boxes, int n, int stride, int conf_idx, double thr,
                                                                  float min_conf, int* __restrict__ keep, int* __restrict__ count,
                                                                  double* __restrict__ iou_out) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) unsigned char nms_smem[];
  NmsPoly* work = reinterpret_cast<NmsPoly*>(nms_smem);                        // [2 * NMS_THREADS]
  double* cx = reinterpret_cast<double*>(work + 2 * NMS_THREADS);               // [n][4]
  double* cy = cx + (size_t)n * 4;                                              // [n][4]
  float* conf = reinterpret_cast<float*>(cy + (size_t)n * 4);                   // [n]
  int* order = reinterpret_cast<int*>(conf + n);                                // [n]
  int* alive = order + n;                                                       // [n]
  __shared__ int kept;
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += NMS_THREADS) {
    const float* b = boxes + (size_t)i * stride;
    const double x = b[0], y = b[1], w = b[2], h = b[3], a = b[4];
    const double c = cos(a), s = sin(a);
    const double lx[4] = {-w, w, w, -w}, ly[4] = {-h, -h, h, h};
#pragma unroll
    for (int k = 0; k < 4; ++k) { cx[i * 4 + k] = c * lx[k] - s * ly[k] + x; cy[i * 4 + k] = s * lx[k] + c * ly[k] + y; }
    conf[i] = b[conf_idx];
    alive[i] = b[conf_idx] > min_conf;  // (model.py:451: boxes at or below the confidence threshold never enter the suppression)
  }
  if (tid == 0) kept = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NMS_THREADS) {  // rank = number of boxes that go before box i
    const float ci = conf[i];
    int r = 0;
    for (int j = 0; j < n; ++j) r += (conf[j] > ci) || (conf[j] == ci && j > i);
    order[r] = i;
  }
  __syncthreads();
  if (iou_out)  // debugging / tests: the whole IoU matrix
    for (int p = tid; p < n * n; p += NMS_THREADS) {
      const int i = p / n, j = p - i * n;
      iou_out[p] = i == j ? 1.0 : nms_iou(cx + i * 4, cy + i * 4, cx + j * 4, cy + j * 4, work + 2 * tid, work + 2 * tid + 1);
    }
  for (int p = 0; p < n; ++p) {
    const int i = order[p];
    if (alive[i]) {  // (uniform: alive[] only changes between the barriers)
      if (tid == 0) keep[kept++] = i;
      for (int q = p + 1 + tid; q < n; q += NMS_THREADS) {
        const int j = order[q];
        if (alive[j] && nms_iou(cx + i * 4, cy + i * 4, cx + j * 4, cy + j * 4, work + 2 * tid, work + 2 * tid + 1) > thr) alive[j] = 0;
      }
    }
    __syncthreads();
  }
  if (tid == 0) *count = kept;
}
}  // namespace

// (B, k, 9) decoded boxes in image pixels -> vehicle coordinates in metres (model.py:447-459 + transfuser_utils.py:388-406): yaw negated,
// origin moved to the ego pixel, axes swapped (image: y front / x right; CARLA: x front / y right), extents swapped, pixels -> metres.
__global__ void bb_image_to_metric_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float ppm, float min_x, float min_y) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* b = in + (size_t)i * 9;
  float* o = out + (size_t)i * 9;
  // box[:2] - np.array([-(min_x * ppm), -(min_y * ppm)]): float32 box, float64 constants -> float64 difference, stored as float32
  const float x = (float)((double)b[0] - (-((double)min_x * (double)ppm)));
  const float y = (float)((double)b[1] - (-((double)min_y * (double)ppm)));
  o[0] = y / ppm; o[1] = x / ppm; o[2] = b[3] / ppm; o[3] = b[2] / ppm;
  o[4] = -b[4];
  o[5] = b[5]; o[6] = b[6]; o[7] = b[7]; o[8] = b[8];
}

extern "C" int tfpp_bb_image_to_metric(const float* boxes, float* out, int n, float pixels_per_meter, float min_x, float min_y, void* stream) {
  if (!boxes || !out || n < 0) return TFPP_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(bb_image_to_metric_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, boxes, out, n, pixels_per_meter, min_x, min_y);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_nms_rotated(const float* boxes, int n, int stride, int conf_idx, double iou_threshold, float min_conf, int32_t* keep,
                                int32_t* count, double* iou_out, void* stream) {
  if (!boxes || !keep || !count || n < 0 || n > NMS_MAX_BOXES || stride < 5 || conf_idx < 0 || conf_idx >= stride) return TFPP_EINVAL;
  const size_t lds = 2 * NMS_THREADS * sizeof(NmsPoly) + (size_t)n * 8 * sizeof(double) + (size_t)n * 3 * sizeof(int) + 64;
  static unsigned long long attr_mask = 0;
  if (tfpp_first_use_on_this_device(&attr_mask)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_rotated_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
  }
  hipLaunchKernelGGL(nms_rotated_kernel, dim3(1), dim3(NMS_THREADS), lds, (hipStream_t)stream, boxes, n, stride, conf_idx, iou_threshold, min_conf, keep, count, iou_out);
  TFPP_CHECK_LAUNCH();
  return 0;
}
