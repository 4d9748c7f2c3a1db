// Camera -> BEV lift of the bev_encoder backbone (team_code/bev_encoder.py:180-201): F.grid_sample of the 32-channel image feature map at
// the pinhole projection of every voxel centre of the 256 x 256 x 96 grid around the car (bilinear, zeros padding, align_corners=False; the
// third grid coordinate is 0 on a depth-1 volume, i.e. exactly the single slice), summed over the 96 heights, divided by the number of visible
// voxels of the column, transposed to image orientation and masked by the visible-BEV-pixel map -- one pass, nothing of the
// (B, 32, 256, 256, 96) intermediate is materialised.  coords: per voxel (d, w, z) the sample position in FEATURE pixels (host constant);
// scale: per output pixel (i = w, j = d) valid / normalizer.  One thread per (b, i, j, 16-byte channel chunk).
#include "common.h"
#include "../../include/tfpp.h"

template <typename T, bool BWD>
__global__ void bev_lift_kernel(const T* __restrict__ feat, const float2* __restrict__ coords, const float* __restrict__ scale, T* __restrict__ out,
                                const T* __restrict__ dout, float* __restrict__ dfeat, int B, int Hf, int Wf, int C, int Dd, int Wd, int Z) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npix = (long)B * Wd * Dd;
  if (t >= npix * CV) return;
  const int cv = (int)(t % CV);
  const long pix = t / CV;
  const int j = (int)(pix % Dd);            // output column = depth index d
  const int i = (int)((pix / Dd) % Wd);     // output row = width index w
  const int b = (int)(pix / ((long)Dd * Wd));
  const float sc = scale[(size_t)i * Dd + j];
  float acc[VEC], g[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (BWD) {
    load_vec<T>(dout + (size_t)pix * C + cv * VEC, g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] *= sc;
  }
  if (sc != 0.f) {
    const float2* cz = coords + ((size_t)j * Wd + i) * Z;
    const T* fb = feat + (size_t)b * Hf * Wf * C + cv * VEC;
    float* db = BWD ? dfeat + (size_t)b * Hf * Wf * C + cv * VEC : nullptr;
    for (int z = 0; z < Z; ++z) {
      const float2 c = cz[z];
      if (!(c.x > -1.f && c.x < (float)Wf && c.y > -1.f && c.y < (float)Hf)) continue;  // all four taps outside (also NaN / inf)
      const float x0f = floorf(c.x), y0f = floorf(c.y);
      const int x0 = (int)x0f, y0 = (int)y0f;
      const float fx = c.x - x0f, fy = c.y - y0f;
#pragma unroll
      for (int tap = 0; tap < 4; ++tap) {
        const int xi = x0 + (tap & 1), yi = y0 + (tap >> 1);
        if (xi < 0 || xi >= Wf || yi < 0 || yi >= Hf) continue;
        const float w = ((tap & 1) ? fx : 1.f - fx) * ((tap >> 1) ? fy : 1.f - fy);
        const size_t o = ((size_t)yi * Wf + xi) * C;
        if (BWD) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) atomicAdd(db + o + e, w * g[e]);
        } else {
          float v[VEC];
          load_vec<T>(fb + o, v);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += w * v[e];
        }
      }
    }
  }
  if (!BWD) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] *= sc;
    store_vec<T>(out + (size_t)pix * C + cv * VEC, acc);
  }
}

extern "C" int tfpp_bev_lift_fwd(const void* feat, const float* coords, const float* scale, void* out, int B, int Hf, int Wf, int C, int D, int W,
                                 int Z, int dtype, void* stream) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (!feat || !coords || !scale || !out || B < 1 || C < VEC || C % VEC || Z < 1) return TFPP_EINVAL;
  const long n = (long)B * W * D * (C / VEC);
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL((bev_lift_kernel<float, false>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)feat, (const float2*)coords, scale, (float*)out,
                       (const float*)nullptr, (float*)nullptr, B, Hf, Wf, C, D, W, Z);
  else
    hipLaunchKernelGGL((bev_lift_kernel<bf16_t, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)feat, (const float2*)coords, scale,
                       (bf16_t*)out, (const bf16_t*)nullptr, (float*)nullptr, B, Hf, Wf, C, D, W, Z);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// adjoint: dfeat (fp32, zeroed by the caller) += lift^T(dout); fp32 atomics (a feature pixel is hit by many voxels of many columns)
extern "C" int tfpp_bev_lift_bwd(const void* dout, const float* coords, const float* scale, float* dfeat, int B, int Hf, int Wf, int C, int D, int W,
                                 int Z, int dtype, void* stream) {
  const int VEC = dtype == TFPP_F32 ? 4 : 8;
  if (!dout || !coords || !scale || !dfeat || B < 1 || C < VEC || C % VEC || Z < 1) return TFPP_EINVAL;
  const long n = (long)B * W * D * (C / VEC);
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL((bev_lift_kernel<float, true>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)nullptr, (const float2*)coords, scale,
                       (float*)nullptr, (const float*)dout, dfeat, B, Hf, Wf, C, D, W, Z);
  else
    hipLaunchKernelGGL((bev_lift_kernel<bf16_t, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, (const float2*)coords, scale,
                       (bf16_t*)nullptr, (const bf16_t*)dout, dfeat, B, Hf, Wf, C, D, W, Z);
  TFPP_CHECK_LAUNCH();
  return 0;
}
