// Device helpers shared by the kernels that consume train-mode BatchNorm statistics straight from the per-M-tile rows of the producing
// convolution (tfpp_bn_rows, include/tfpp.h): bn_rows_kernels.hip (elementwise passes) and conv3x3_halo.hip (normalise-on-load).
// All of them assume 256-thread workgroups.
#pragma once
#include "common.h"
#include "../../include/tfpp.h"

// Column sums of the rows [nrows][2C] (first half | second half) for the NCH = cb * VEC channels of this workgroup's block, in double, fixed
// order.  Returns true in the threads t < NCH whose channel exists; they hold the two totals.  sm: 2 * 256 doubles of LDS.  Contains barriers.
template <int U = 24>
__device__ __forceinline__ bool rows_block_sum(const float* __restrict__ partial, int nrows, int C, int c_base, int nch, double* sm, int& c,
                                               double& s0, double& s1) {
  // U rows (2 U loads) per thread and trip, ALL issued before the first use: the rows were written by the kernel in front, i.e. they come from
  // the memory side (~2 us per dependent round trip in the captured step).  The first version unrolled by 4: six round trips for the 96 rows
  // of a stage-3 layer, and the "free" prologue cost as much as the finalize launch it replaced (profiles/r06_*).  With U = 24 and 4 row
  // groups (64 channels per workgroup) 96 rows are one trip; callers with fewer channels per workgroup (more row groups) pass a smaller U.
  const int t = threadIdx.x;
  const int rg_n = 256 / nch;  // row groups (>= 2: nch <= 96)
  const int col = t % nch, rg = t / nch;
  c = c_base + col;
  double a = 0.0, b = 0.0;
  if (rg < rg_n && c < C) {
    const size_t pitch = (size_t)2 * C;
    for (int k0 = rg; k0 < nrows; k0 += U * rg_n) {
      float va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * rg_n;
        const float* row = partial + (size_t)(k < nrows ? k : rg) * pitch;  // (rows past the end re-read a valid one; not added)
        va[u] = row[c];
        vb[u] = row[C + c];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (k0 + u * rg_n < nrows) { a += (double)va[u]; b += (double)vb[u]; }
    }
  }
  sm[t] = a;
  sm[256 + t] = b;
  __syncthreads();
  if (t >= nch || c >= C) return false;
  s0 = 0.0;
  s1 = 0.0;
  for (int r = 0; r < rg_n; ++r) { s0 += sm[r * nch + t]; s1 += sm[256 + r * nch + t]; }
  return true;
}

// Forward statistics of this workgroup's channel block into LDS (sc_s / sh_s: nch floats each).  bn.partial != NULL: finalize from the rows
// (what bn_finalize_partials_kernel computes, same formulas); `writer` workgroups also store the results / update the running statistics.
// Otherwise read the final scale / shift.  Contains barriers; every thread of the workgroup must call it.
template <int U = 24>
__device__ __forceinline__ void bn_block_scale_shift(const tfpp_bn_rows& bn, int c_base, int nch, bool writer, bool first_writer, double* sm,
                                                     float* sc_s, float* sh_s) {
  const int t = threadIdx.x;
  if (bn.partial) {
    int c;
    double s0, s1;
    if (rows_block_sum<U>(bn.partial, bn.nrows, bn.C, c_base, nch, sm, c, s0, s1)) {
      const double n = (double)bn.count;
      const double m = s0 / n;
      double var = s1 / n - m * m;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)bn.eps));
      const float g = bn.gamma ? bn.gamma[c] : 1.f, b = bn.beta ? bn.beta[c] : 0.f;
      const float sc = g * invstd, sh = b - (float)m * g * invstd;
      sc_s[t] = sc;
      sh_s[t] = sh;
      if (writer) {
        bn.scale[c] = sc;
        bn.shift[c] = sh;
        if (bn.save_mean) bn.save_mean[c] = (float)m;
        if (bn.save_invstd) bn.save_invstd[c] = invstd;
        if (bn.running_mean) bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * (float)m;
        if (bn.running_var) {
          const double unb = bn.count > 1 ? var * n / (n - 1.0) : var;
          bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * (float)unb;
        }
      }
    }
    if (first_writer && t == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;
  } else if (t < nch && c_base + t < bn.C) {
    sc_s[t] = bn.scale[c_base + t];
    sh_s[t] = bn.shift[c_base + t];
  }
  __syncthreads();
}

