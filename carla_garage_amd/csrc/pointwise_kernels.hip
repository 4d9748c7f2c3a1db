// HBM-bound kernels: weight packing, layout changes, pooling / bilinear resampling, elementwise ops, squeeze-excite.
// All activations are NHWC; every kernel moves 16-byte vectors per lane along the channel dimension.
#include "common.h"
#include <cstdlib>
#include "../../include/tfpp.h"

#define PW_THREADS 256
static inline dim3 grid1d(long n, int per = PW_THREADS) {
  long b = (n + per - 1) / per;
  return dim3((unsigned)(b < 1 ? 1 : b));
}

// ---------------------------------------------------------------------------------------------------------------
// packing
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int cin_g, int R, int S, int G,
                                        int ks_pad, int n_pad, int transpose, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n_g = Cout / G, RS = R * S;
  float v = 0.f;
  if (!transpose) {  // [G][n_pad][RS*ks_pad]
    const int K = RS * ks_pad;
    const int k = (int)(i % K);
    const long t = i / K;
    const int n = (int)(t % n_pad), g = (int)(t / n_pad);
    const int rs = k / ks_pad, c = k - rs * ks_pad;
    if (n < n_g && c < cin_g) v = w[((size_t)(g * n_g + n) * cin_g + c) * RS + rs];
  } else {  // [G][cin_g][RS*n_pad]
    const int K = RS * n_pad;
    const int k = (int)(i % K);
    const long t = i / K;
    const int c = (int)(t % cin_g), g = (int)(t / cin_g);
    const int rs = k / n_pad, n = k - rs * n_pad;
    if (n < n_g) v = w[((size_t)(g * n_g + n) * cin_g + c) * RS + rs];
  }
  out[i] = ElemTraits<T>::from_f(v);
}

extern "C" int tfpp_pack_conv_weight(const float* w, void* out, int Cout, int cin_g, int R, int S, int G, int ks_pad, int n_pad,
                                     int transpose, int dtype, void* stream) {
  if (!w || !out || G < 1 || Cout % G) return TFPP_EINVAL;
  const long total = transpose ? (long)G * cin_g * R * S * n_pad : (long)G * n_pad * R * S * ks_pad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, grid1d(total), dim3(PW_THREADS), 0, st, w, (float*)out, Cout, cin_g, R, S, G,
                       ks_pad, n_pad, transpose, total);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<bf16_t>, grid1d(total), dim3(PW_THREADS), 0, st, w, (bf16_t*)out, Cout, cin_g, R, S, G,
                       ks_pad, n_pad, transpose, total);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// out[r][c] = in[rmap(r)][cmap(c)]  (transpose_in: in[cmap(c)][rmap(r)]); -1 in a map -> 0
template <typename T>
__global__ void pack2d_kernel(const float* __restrict__ in, T* __restrict__ out, const int* __restrict__ row_map,
                              const int* __restrict__ col_map, int rows_out, int cols_out, long in_ld, long out_ld, int transpose_in) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows_out * cols_out) return;
  const int r = (int)(i / cols_out), c = (int)(i - (long)r * cols_out);
  const int sr = row_map ? row_map[r] : r, sc = col_map ? col_map[c] : c;
  float v = 0.f;
  if (sr >= 0 && sc >= 0) v = transpose_in ? in[(size_t)sc * in_ld + sr] : in[(size_t)sr * in_ld + sc];
  out[(size_t)r * out_ld + c] = ElemTraits<T>::from_f(v);
}

extern "C" int tfpp_pack2d(const float* in, void* out, const int* row_map, const int* col_map, int rows_out, int cols_out,
                           int64_t in_ld, int64_t out_ld, int transpose_in, int dtype, void* stream) {
  if (!in || !out) return TFPP_EINVAL;
  const long total = (long)rows_out * cols_out;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32)
    hipLaunchKernelGGL(pack2d_kernel<float>, grid1d(total), dim3(PW_THREADS), 0, st, in, (float*)out, row_map, col_map, rows_out, cols_out,
                       (long)in_ld, (long)out_ld, transpose_in);
  else
    hipLaunchKernelGGL(pack2d_kernel<bf16_t>, grid1d(total), dim3(PW_THREADS), 0, st, in, (bf16_t*)out, row_map, col_map, rows_out,
                       cols_out, (long)in_ld, (long)out_ld, transpose_in);
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <typename TI, typename TO> __global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = ElemTraits<TO>::from_f(ElemTraits<TI>::to_f(in[i]));
}

extern "C" int tfpp_cast(const void* in, void* out, int64_t n, int dtype_in, int dtype_out, void* stream) {
  if (!in || !out) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long blocks = (n + PW_THREADS - 1) / PW_THREADS;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  dim3 g((unsigned)blocks), b(PW_THREADS);
  if (dtype_in == TFPP_F32 && dtype_out == TFPP_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, (const float*)in, (bf16_t*)out, (long)n);
  else if (dtype_in == TFPP_BF16 && dtype_out == TFPP_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)in, (float*)out, (long)n);
  else if (dtype_in == TFPP_F32 && dtype_out == TFPP_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, (const float*)in, (float*)out, (long)n);
  else if (dtype_in == TFPP_BF16 && dtype_out == TFPP_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)in, (bf16_t*)out, (long)n);
  else return TFPP_EINVAL;
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Narrow host dtypes of the reference's collated batch (uint8 camera frames and semantic maps, int32 detection targets: team_code/data.py:511-522,
// 725-728) widened on the GPU to what train.py:688-766 asks ``.to(device, dtype=...)`` for, so that PCIe carries 1 or 4 bytes per value instead of
// 4 or 8.  16 source bytes per thread.
template <typename S, typename D>
__global__ void widen_kernel(const S* __restrict__ in, D* __restrict__ out, long nvec, long n) {
  constexpr int V = 16 / sizeof(S);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nvec) {
    const uint4 raw = reinterpret_cast<const uint4*>(in)[i];
    const S* v = reinterpret_cast<const S*>(&raw);
    D r[V];
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = (D)v[k];
    constexpr int Q = V * sizeof(D) / 16;
    uint4* o = reinterpret_cast<uint4*>(out) + i * Q;
#pragma unroll
    for (int k = 0; k < Q; ++k) o[k] = reinterpret_cast<const uint4*>(r)[k];
  } else if (i == nvec) {
    for (long j = nvec * V; j < n; ++j) out[j] = (D)in[j];
  }
}

template <typename S, typename D>
static int launch_widen(const void* in, void* out, int64_t n, hipStream_t st) {
  const long nvec = n / (16 / (long)sizeof(S));
  hipLaunchKernelGGL((widen_kernel<S, D>), dim3((unsigned)((nvec + 1 + 255) / 256)), dim3(256), 0, st, (const S*)in, (D*)out, nvec, (long)n);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_widen(const void* in, void* out, int64_t n, int src_kind, int dst_kind, void* stream) {
  if (!in || !out || n < 0 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (src_kind == 0 && dst_kind == 0) return launch_widen<unsigned char, float>(in, out, n, st);
  if (src_kind == 0 && dst_kind == 1) return launch_widen<unsigned char, long long>(in, out, n, st);
  if (src_kind == 1 && dst_kind == 0) return launch_widen<int, float>(in, out, n, st);
  if (src_kind == 1 && dst_kind == 1) return launch_widen<int, long long>(in, out, n, st);
  return TFPP_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------
// layout changes at the boundary
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_affine_kernel(const float* __restrict__ in, T* __restrict__ out, const float* __restrict__ mul,
                                           const float* __restrict__ add, int B, int C, int HW, int cpad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one pixel
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), pix = (int)(i - (long)b * HW);
  T* o = out + (size_t)i * cpad;
  for (int c0 = 0; c0 < cpad; c0 += ElemTraits<T>::VEC) {
    float v[ElemTraits<T>::VEC];
#pragma unroll
    for (int e = 0; e < ElemTraits<T>::VEC; ++e) {
      const int c = c0 + e;
      float x = 0.f;
      if (c < C) {
        x = in[((size_t)b * C + c) * HW + pix];
        if (mul) x = x * mul[c] + add[c];
      }
      v[e] = x;
    }
    store_vec<T>(o + c0, v);
  }
}

extern "C" int tfpp_nchw_to_nhwc_affine(const float* in, void* out, const float* mul, const float* add, int B, int C, int H, int W,
                                        int cpad, int dtype, void* stream) {
  if (!in || !out || cpad < C) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * H * W;
  if (dtype == TFPP_F32) {
    if (cpad % 4) return TFPP_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_affine_kernel<float>, grid1d(n), dim3(PW_THREADS), 0, st, in, (float*)out, mul, add, B, C, H * W, cpad);
  } else {
    if (cpad % 8) return TFPP_EINVAL;
    hipLaunchKernelGGL(nchw_to_nhwc_affine_kernel<bf16_t>, grid1d(n), dim3(PW_THREADS), 0, st, in, (bf16_t*)out, mul, add, B, C, H * W, cpad);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// The camera frame as the caller holds it: uint8, either HWC (what cv2.imdecode returns, sensor_agent.py:277-286: BGR, swap = 1 turns it into
// the RGB order the network was trained on) or CHW (the collated loader batch before train.py:750's .to(float32)).  Fused: channel swap +
// uint8 -> float + normalize_imagenet + NHWC + zero channel padding -- the host no longer transposes or widens, the upload is 4x smaller.
template <typename T, bool HWC>
__global__ void u8_to_nhwc_affine_kernel(const unsigned char* __restrict__ in, T* __restrict__ out, const float* __restrict__ mul,
                                         const float* __restrict__ add, int B, int C, int HW, int cpad, int swap) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // one pixel
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), pix = (int)(i - (long)b * HW);
  T* o = out + (size_t)i * cpad;
  for (int c0 = 0; c0 < cpad; c0 += ElemTraits<T>::VEC) {
    float v[ElemTraits<T>::VEC];
#pragma unroll
    for (int e = 0; e < ElemTraits<T>::VEC; ++e) {
      const int c = c0 + e;
      float x = 0.f;
      if (c < C) {
        const int cs = swap ? C - 1 - c : c;  // source channel
        x = (float)(HWC ? in[(size_t)i * C + cs] : in[((size_t)b * C + cs) * HW + pix]);
        if (mul) x = x * mul[c] + add[c];
      }
      v[e] = x;
    }
    store_vec<T>(o + c0, v);
  }
}

extern "C" int tfpp_u8_to_nhwc_affine(const uint8_t* in, void* out, const float* mul, const float* add, int B, int C, int H, int W, int cpad,
                                      int hwc, int swap, int dtype, void* stream) {
  if (!in || !out || cpad < C || C < 1) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * H * W;
  if (cpad % (dtype == TFPP_F32 ? 4 : 8)) return TFPP_EINVAL;
#define U8_LAUNCH(TT, L) hipLaunchKernelGGL((u8_to_nhwc_affine_kernel<TT, L>), grid1d(n), dim3(PW_THREADS), 0, st, in, (TT*)out, mul, add, B, C, H * W, cpad, swap)
  if (dtype == TFPP_F32) { if (hwc) U8_LAUNCH(float, true); else U8_LAUNCH(float, false); }
  else { if (hwc) U8_LAUNCH(bf16_t, true); else U8_LAUNCH(bf16_t, false); }
#undef U8_LAUNCH
  TFPP_CHECK_LAUNCH();
  return 0;
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int C, int HW, long in_ld, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C * HW) return;
  const int pix = (int)(i % HW);
  const long t = i / HW;
  const int c = (int)(t % C), b = (int)(t / C);
  out[i] = apply_act(ElemTraits<T>::to_f(in[((size_t)b * HW + pix) * in_ld + c]), act);
}

extern "C" int tfpp_nhwc_to_nchw(const void* in, float* out, int B, int C, int H, int W, int64_t in_ld, int act, int dtype,
                                 void* stream) {
  if (!in || !out) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * C * H * W;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid1d(n), dim3(PW_THREADS), 0, st, (const float*)in, out, B, C, H * W, (long)in_ld, act);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid1d(n), dim3(PW_THREADS), 0, st, (const bf16_t*)in, out, B, C, H * W, (long)in_ld, act);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// gradient counterpart: NCHW fp32 (caller's dL/dpred) -> NHWC dtype with channel stride out_ld, padding channels zero
template <typename T>
__global__ void nchw_to_nhwc_grad_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int HW, long out_ld) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW * out_ld) return;
  const int c = (int)(i % out_ld);
  const long bp = i / out_ld;
  const int pix = (int)(bp % HW), b = (int)(bp / HW);
  out[i] = ElemTraits<T>::from_f(c < C ? in[((size_t)b * C + c) * HW + pix] : 0.f);
}

extern "C" int tfpp_nchw_to_nhwc_pad(const float* in, void* out, int B, int C, int H, int W, int64_t out_ld, int dtype, void* stream) {
  if (!in || !out || out_ld < C) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * H * W * out_ld;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(nchw_to_nhwc_grad_kernel<float>, grid1d(n), dim3(PW_THREADS), 0, st, in, (float*)out, B, C, H * W, (long)out_ld);
  else hipLaunchKernelGGL(nchw_to_nhwc_grad_kernel<bf16_t>, grid1d(n), dim3(PW_THREADS), 0, st, in, (bf16_t*)out, B, C, H * W, (long)out_ld);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// elementwise over [rows, C] with 16-byte vectors
// ---------------------------------------------------------------------------------------------------------------
// y = act(x * gate[b,c] * scale[c] + shift[c] + res); column-fixed layout, scale/shift live in registers.
// ACT < 0: activation chosen at run time (uniform switch); otherwise compiled in.
template <typename T, bool RES, bool GATE, int ACT>
__global__ void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                  const T* __restrict__ res, const float* __restrict__ gate, T* __restrict__ y, long rows, int CV,
                                  int sw, int rp, long rows_per_batch, int act_rt) {
  constexpr int VEC = ElemTraits<T>::VEC;
  int rr, cv;
  if (!col_thread(sw, rp, CV, rr, cv)) return;
  const int c0 = cv * VEC, C = CV * VEC;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) { sc[e] = scale ? scale[c0 + e] : 1.f; sh[e] = shift ? shift[c0 + e] : 0.f; }
  const long stride = (long)gridDim.x * rp;
  constexpr int U = 2;
  auto body = [&](long r, const uint4& xv, const uint4& rv) {
    float v[VEC], q[VEC];
    unpack16<T>(xv, v);
    if (RES) unpack16<T>(rv, q);
    if (GATE) {
      const float4* gp = reinterpret_cast<const float4*>(gate + (size_t)((unsigned)r / (unsigned)rows_per_batch) * C + c0);
#pragma unroll
      for (int h = 0; h < VEC / 4; ++h) {
        const float4 g4 = gp[h];
        v[4 * h + 0] *= g4.x; v[4 * h + 1] *= g4.y; v[4 * h + 2] *= g4.z; v[4 * h + 3] *= g4.w;
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = v[e] * sc[e] + sh[e];
      if (RES) t += q[e];
      v[e] = ACT < 0 ? apply_act(t, act_rt) : apply_act(t, ACT);
    }
    store_vec<T>(y + (size_t)r * C + c0, v);
  };
  long r = (long)blockIdx.x * rp + rr;
  for (; r + (U - 1) * stride < rows; r += U * stride) {
    uint4 xv[U], rv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (size_t)(r + u * stride) * C + c0;
      xv[u] = *reinterpret_cast<const uint4*>(x + off);
      if (RES) rv[u] = *reinterpret_cast<const uint4*>(res + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) body(r + u * stride, xv[u], rv[u]);
  }
  for (; r < rows; r += stride) {
    const size_t off = (size_t)r * C + c0;
    uint4 xv = *reinterpret_cast<const uint4*>(x + off), rv = make_uint4(0, 0, 0, 0);
    if (RES) rv = *reinterpret_cast<const uint4*>(res + off);
    body(r, xv, rv);
  }
}

template <typename T>
static int launch_affine_act(const void* x, const float* scale, const float* shift, const void* res, const float* gate, void* y, long rows,
                             int C, long rows_per_batch, int act, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  const int CV = C / VEC;
  const ColLayout l = col_layout(CV);
  dim3 grid((unsigned)col_blocks_x(rows, l, 4, 1 << 20), (unsigned)l.ny);
#define AA(R_, G_, A_) hipLaunchKernelGGL((affine_act_kernel<T, R_, G_, A_>), grid, dim3(256), 0, st, (const T*)x, scale, shift, (const T*)res, gate, (T*)y, rows, CV, l.sw, l.rp, rows_per_batch, act)
#define AA_ACT(R_, G_) do { if (act == ACT_RELU) AA(R_, G_, ACT_RELU); else if (act == ACT_NONE) AA(R_, G_, ACT_NONE); else AA(R_, G_, -1); } while (0)
  if (res) { if (gate) AA_ACT(true, true); else AA_ACT(true, false); }
  else { if (gate) AA_ACT(false, true); else AA_ACT(false, false); }
#undef AA_ACT
#undef AA
  TFPP_CHECK_LAUNCH();
  return 0;
}

static inline dim3 grid_stride(long n) {
  long b = (n + PW_THREADS - 1) / PW_THREADS;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

extern "C" int tfpp_affine_act(const void* x, const float* scale, const float* shift, const void* res, const float* gate, void* y,
                               int64_t rows, int C, int64_t rows_per_batch, int act, int dtype, void* stream) {
  if (!x || !y || (gate && rows_per_batch < 1)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_affine_act<float>(x, scale, shift, res, gate, y, (long)rows, C, (long)rows_per_batch, act, st)
                           : launch_affine_act<bf16_t>(x, scale, shift, res, gate, y, (long)rows, C, (long)rows_per_batch, act, st);
}

// y = a + dropout(b)
template <typename T>
__global__ void add_dropout_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n, float p, float inv_keep,
                                   unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;  // per-step device counter: hipGraph replays draw fresh masks
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float v = ElemTraits<T>::to_f(b[i]);
    if (p > 0.f) v *= dropout_scale(seed, (unsigned long long)i, p, inv_keep);
    if (a) v += ElemTraits<T>::to_f(a[i]);
    y[i] = ElemTraits<T>::from_f(v);
  }
}

extern "C" int tfpp_add_dropout(const void* a, const void* b, void* y, int64_t n, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype, void* stream) {
  if (!b || !y) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(add_dropout_kernel<float>, grid_stride(n), dim3(PW_THREADS), 0, st, (const float*)a, (const float*)b, (float*)y, (long)n, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  else hipLaunchKernelGGL(add_dropout_kernel<bf16_t>, grid_stride(n), dim3(PW_THREADS), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, (long)n, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// y = x + bcast[i % period]
template <typename T>
__global__ void add_bcast_kernel(const T* __restrict__ x, const float* __restrict__ bc, T* __restrict__ y, long n, long period) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(x[i]) + bc[i % period]);
}

extern "C" int tfpp_add_bcast(const void* x, const float* bcast, void* y, int64_t n, int64_t period, int dtype, void* stream) {
  if (!x || !y || !bcast || period < 1) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(add_bcast_kernel<float>, grid_stride(n), dim3(PW_THREADS), 0, st, (const float*)x, bcast, (float*)y, (long)n, (long)period);
  else hipLaunchKernelGGL(add_bcast_kernel<bf16_t>, grid_stride(n), dim3(PW_THREADS), 0, st, (const bf16_t*)x, bcast, (bf16_t*)y, (long)n, (long)period);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dx = dy * act'(.)   (y = activation output for relu/sigmoid/tanh, activation input for gelu)
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long n, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float g = ElemTraits<T>::to_f(dy[i]), v = ElemTraits<T>::to_f(y[i]);
    float d;
    switch (act) {
      case ACT_RELU: d = v > 0.f ? g : 0.f; break;
      case ACT_SIGMOID: d = g * v * (1.f - v); break;
      case ACT_TANH: d = g * (1.f - v * v); break;
      case ACT_GELU: d = g * (0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v)); break;
      default: d = g;
    }
    dx[i] = ElemTraits<T>::from_f(d);
  }
}

extern "C" int tfpp_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, int dtype, void* stream) {
  if (!dy || !y || !dx) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, grid_stride(n), dim3(PW_THREADS), 0, st, (const float*)dy, (const float*)y, (float*)dx, (long)n, act);
  else hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid_stride(n), dim3(PW_THREADS), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, (long)n, act);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// y += a * x
template <typename T> __global__ void axpy_kernel(const T* __restrict__ x, T* __restrict__ y, long n, float a) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(y[i]) + a * ElemTraits<T>::to_f(x[i]));
}
extern "C" int tfpp_axpy(const void* x, void* y, int64_t n, float a, int dtype, void* stream) {
  if (!x || !y) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(axpy_kernel<float>, grid_stride(n), dim3(PW_THREADS), 0, st, (const float*)x, (float*)y, (long)n, a);
  else hipLaunchKernelGGL(axpy_kernel<bf16_t>, grid_stride(n), dim3(PW_THREADS), 0, st, (const bf16_t*)x, (bf16_t*)y, (long)n, a);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// y[b, pix, c] = x[b, pix, c] * m[pix]  (visibility mask, model.py:385); x,y have channel stride ld
template <typename T>
__global__ void mul_pixmask_kernel(const T* __restrict__ x, const float* __restrict__ m, T* __restrict__ y, long n, long ld, long HW) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(x[i]) * m[(i / ld) % HW]);
}
extern "C" int tfpp_mul_pixmask(const void* x, const float* m, void* y, int64_t n, int64_t ld, int64_t HW, int dtype, void* stream) {
  if (!x || !y || !m) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) hipLaunchKernelGGL(mul_pixmask_kernel<float>, grid_stride(n), dim3(PW_THREADS), 0, st, (const float*)x, m, (float*)y, (long)n, (long)ld, (long)HW);
  else hipLaunchKernelGGL(mul_pixmask_kernel<bf16_t>, grid_stride(n), dim3(PW_THREADS), 0, st, (const bf16_t*)x, m, (bf16_t*)y, (long)n, (long)ld, (long)HW);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// column sums: out[c] += sum_rows x[row*ld + c]   (bias gradients); block = 64 row-slots x 4... generic layout below
// ---------------------------------------------------------------------------------------------------------------
// Second stage of the two-stage column reductions: out[b][c] (+)= mul * sum_{k<P} partial[(b*P + k)*Cw + c].
// block = 64 columns x 4 partial slots (coalesced 256-byte row segments, <= P/4 independent loads per thread).
__global__ void col_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, int Cw, float mul, int accumulate) {
  const int cl = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
  float s = 0.f;
  if (c < Cw) {
    const float* base = partial + (size_t)b * P * Cw + c;
#pragma unroll 8
    for (int k = ks; k < P; k += 4) s += base[(size_t)k * Cw];
  }
  __shared__ float sm[4][64];
  sm[ks][cl] = s;
  __syncthreads();
  if (ks == 0 && c < Cw) {
    const float t = (sm[0][cl] + sm[1][cl] + sm[2][cl] + sm[3][cl]) * mul;
    float* o = out + (size_t)b * Cw + c;
    *o = accumulate ? *o + t : t;
  }
}
#define COLSUM_MAX_PARTIALS 128
#define HW_MAX_PARTIALS 64

// Generic path (any C / ld): 64 columns x 4 row slots, scalar loads.
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int C, long ld, long rows_per_block) {
  const int cl = threadIdx.x & 63, rs = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rs; r < r1; r += 4) s += ElemTraits<T>::to_f(x[(size_t)r * ld + c]);
  __shared__ float sm[4][64];
  sm[rs][cl] = s;
  __syncthreads();
  if (rs == 0 && c < C) atomicAdd(out + c, sm[0][cl] + sm[1][cl] + sm[2][cl] + sm[3][cl]);
}

// Vector path (C % VEC == 0, ld % VEC == 0): column-fixed layout, 16-byte loads, 4 in flight per lane.
template <typename T>
__global__ void colsum_vec_kernel(const T* __restrict__ x, float* __restrict__ partial, long rows, int CV, long ld, int sw, int rp) {
  constexpr int VEC = ElemTraits<T>::VEC;
  int rr, cv;
  const bool active = col_thread(sw, rp, CV, rr, cv);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (active) {
    const long stride = (long)gridDim.x * rp;
    constexpr int U = 4;
    long r = (long)blockIdx.x * rp + rr;
    for (; r + (U - 1) * stride < rows; r += U * stride) {
      uint4 xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const uint4*>(x + (size_t)(r + u * stride) * ld + cv * VEC);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[VEC];
        unpack16<T>(xv[u], v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[e];
      }
    }
    for (; r < rows; r += stride) {
      float v[VEC];
      load_vec<T>(x + (size_t)r * ld + cv * VEC, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += v[e];
    }
  }
  __shared__ float sm[VEC * 256];
  col_block_reduce<VEC>(acc, sw, rp, rr, sm);
  if (active && rr == 0) {
    float* o = partial + (size_t)blockIdx.x * CV * VEC + cv * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = acc[e];
  }
}

template <typename T> static int launch_colsum(const void* x, float* out, float* scratch, long rows, int C, long ld, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (scratch && C % VEC == 0 && ld % VEC == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const ColLayout l = col_layout(C / VEC);
    int nb = col_blocks_x(rows, l, 8, 2048);
    if (nb > COLSUM_MAX_PARTIALS) nb = COLSUM_MAX_PARTIALS;
    hipLaunchKernelGGL(colsum_vec_kernel<T>, dim3((unsigned)nb, (unsigned)l.ny), dim3(256), 0, st, (const T*)x, scratch, rows, C / VEC, ld, l.sw, l.rp);
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)((C + 63) / 64), 1), dim3(256), 0, st, scratch, out, nb, C, 1.f, 1);
  } else {
    // rows per workgroup: keep <= ~128 atomics per output address while still filling the chip for narrow tensors
    long rpb = 256;
    const long col_blocks = (C + 63) / 64;
    while ((rows + rpb - 1) / rpb > 128 && (rows + rpb - 1) / rpb * col_blocks > 2048) rpb *= 2;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((C + 63) / 64));
    if (grid.x < 1) grid.x = 1;
    hipLaunchKernelGGL(colsum_kernel<T>, grid, dim3(256), 0, st, (const T*)x, out, rows, C, ld, rpb);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_colsum(const void* x, float* out, float* scratch, int64_t rows, int C, int64_t ld, int dtype, void* stream) {
  if (!x || !out) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_colsum<float>(x, out, scratch, (long)rows, C, (long)ld, st)
                           : launch_colsum<bf16_t>(x, out, scratch, (long)rows, C, (long)ld, st);
}

extern "C" int tfpp_reduce_scratch_floats(int B, int C) {
  const long hw = (long)B * HW_MAX_PARTIALS * C, cs = (long)COLSUM_MAX_PARTIALS * C;
  return (int)(hw > cs ? hw : cs);
}

// ---------------------------------------------------------------------------------------------------------------
// pooling / resampling
// ---------------------------------------------------------------------------------------------------------------
// adaptive average pooling with uniform windows (H % Ho == 0, W % Wo == 0): thread per (b,ho,wo,channel-vector)
template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo, long y_ld) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(i % CV);
  long t = i / CV;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int kh = H / Ho, kw = W / Wo;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int dh = 0; dh < kh; ++dh)
    for (int dw = 0; dw < kw; ++dw) {
      float v[VEC];
      load_vec<T>(x + ((size_t)(b * H + ho * kh + dh) * W + wo * kw + dw) * C + cv * VEC, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += v[e];
    }
  const float inv = 1.f / (float)(kh * kw);
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] *= inv;
  store_vec<T>(y + ((size_t)(b * Ho + ho) * Wo + wo) * y_ld + cv * VEC, acc);
}

extern "C" int tfpp_avgpool_fwd(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t y_ld, int dtype, void* stream) {
  if (!x || !y || H % Ho || W % Wo) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) {
    if (C % 4 || y_ld % 4) return TFPP_EINVAL;
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, grid1d((long)B * Ho * Wo * (C / 4)), dim3(PW_THREADS), 0, st, (const float*)x, (float*)y, B, H, W, C, Ho, Wo, (long)y_ld);
  } else {
    if (C % 8 || y_ld % 8) return TFPP_EINVAL;
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, grid1d((long)B * Ho * Wo * (C / 8)), dim3(PW_THREADS), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, Ho, Wo, (long)y_ld);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dx[b,h,w,:] += dy[b,h/kh,w/kw,:] / (kh*kw)
template <typename T>
__global__ void avgpool_bwd_add_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo, long dy_ld) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * H * W * CV) return;
  const int cv = (int)(i % CV);
  long t = i / CV;
  const int w = (int)(t % W); t /= W;
  const int h = (int)(t % H);
  const int b = (int)(t / H);
  const int kh = H / Ho, kw = W / Wo;
  float g[VEC], v[VEC];
  load_vec<T>(dy + ((size_t)(b * Ho + h / kh) * Wo + w / kw) * dy_ld + cv * VEC, g);
  T* d = dx + ((size_t)(b * H + h) * W + w) * C + cv * VEC;
  load_vec<T>(d, v);
  const float inv = 1.f / (float)(kh * kw);
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] += g[e] * inv;
  store_vec<T>(d, v);
}

extern "C" int tfpp_avgpool_bwd_add(const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t dy_ld, int dtype, void* stream) {
  if (!dy || !dx || H % Ho || W % Wo) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) {
    if (C % 4 || dy_ld % 4) return TFPP_EINVAL;
    hipLaunchKernelGGL(avgpool_bwd_add_kernel<float>, grid1d((long)B * H * W * (C / 4)), dim3(PW_THREADS), 0, st, (const float*)dy, (float*)dx, B, H, W, C, Ho, Wo, (long)dy_ld);
  } else {
    if (C % 8 || dy_ld % 8) return TFPP_EINVAL;
    hipLaunchKernelGGL(avgpool_bwd_add_kernel<bf16_t>, grid1d((long)B * H * W * (C / 8)), dim3(PW_THREADS), 0, st, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C, Ho, Wo, (long)dy_ld);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// PyTorch bilinear, align_corners=False: src = max(0, (o+0.5)*in/out - 0.5); i0=floor(src); i1=min(i0+1,in-1); l=src-i0
__device__ __forceinline__ void bilin_coord(int o, int in_sz, int out_sz, int& i0, int& i1, float& l1) {
  const float scale = (float)in_sz / (float)out_sz;
  float src = ((float)o + 0.5f) * scale - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_sz - 1) i0 = in_sz - 1;
  i1 = i0 + ((i0 < in_sz - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

// y = base + bilinear(x) [* mul[pixel]] ; output NHWC dtype (ld y_ld) or NCHW fp32
template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, const T* __restrict__ base, const float* __restrict__ mul, void* __restrict__ yv,
                                    int B, int Hi, int Wi, int Ho, int Wo, int C, long x_ld, long y_ld, int out_nchw_f32, int c_real) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Ho * Wo * CV) return;
  const int cv = (int)(i % CV);
  long t = i / CV;
  const int wo = (int)(t % Wo); t /= Wo;
  const int ho = (int)(t % Ho);
  const int b = (int)(t / Ho);
  int h0, h1, w0, w1;
  float lh, lw;
  bilin_coord(ho, Hi, Ho, h0, h1, lh);
  bilin_coord(wo, Wi, Wo, w0, w1, lw);
  float v00[VEC], v01[VEC], v10[VEC], v11[VEC], o[VEC];
  const T* xb = x + (size_t)b * Hi * Wi * x_ld + cv * VEC;
  load_vec<T>(xb + ((size_t)h0 * Wi + w0) * x_ld, v00);
  load_vec<T>(xb + ((size_t)h0 * Wi + w1) * x_ld, v01);
  load_vec<T>(xb + ((size_t)h1 * Wi + w0) * x_ld, v10);
  load_vec<T>(xb + ((size_t)h1 * Wi + w1) * x_ld, v11);
  const float m = mul ? mul[(size_t)ho * Wo + wo] : 1.f;
#pragma unroll
  for (int e = 0; e < VEC; ++e)
    o[e] = ((1.f - lh) * ((1.f - lw) * v00[e] + lw * v01[e]) + lh * ((1.f - lw) * v10[e] + lw * v11[e])) * m;
  const size_t pix = ((size_t)b * Ho + ho) * Wo + wo;
  if (base) {
    float bb[VEC];
    load_vec<T>(base + pix * y_ld + cv * VEC, bb);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] += bb[e];
  }
  if (out_nchw_f32) {
    float* y = reinterpret_cast<float*>(yv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = cv * VEC + e;
      if (c < c_real) y[((size_t)b * c_real + c) * Ho * Wo + (size_t)ho * Wo + wo] = o[e];
    }
  } else {
    store_vec<T>(reinterpret_cast<T*>(yv) + pix * y_ld + cv * VEC, o);
  }
}

extern "C" int tfpp_bilinear_fwd(const void* x, const void* base, const float* mul, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                                 int64_t x_ld, int64_t y_ld, int out_nchw_f32, int c_real, int dtype, void* stream) {
  if (!x || !y) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) {
    if (C % 4 || x_ld % 4 || (!out_nchw_f32 && y_ld % 4)) return TFPP_EINVAL;
    hipLaunchKernelGGL(bilinear_fwd_kernel<float>, grid1d((long)B * Ho * Wo * (C / 4)), dim3(PW_THREADS), 0, st, (const float*)x, (const float*)base, mul, y, B, Hi, Wi, Ho, Wo, C, (long)x_ld, (long)y_ld, out_nchw_f32, c_real);
  } else {
    if (C % 8 || x_ld % 8 || (!out_nchw_f32 && y_ld % 8)) return TFPP_EINVAL;
    hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, grid1d((long)B * Ho * Wo * (C / 8)), dim3(PW_THREADS), 0, st, (const bf16_t*)x, (const bf16_t*)base, mul, y, B, Hi, Wi, Ho, Wo, C, (long)x_ld, (long)y_ld, out_nchw_f32, c_real);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dx[b,hi,wi,:] = sum over output pixels (ho,wo) whose stencil touches (hi,wi) of weight * dy[b,ho,wo,:] * mul[ho,wo]
// acc[e] of the RS consecutive lanes of a group summed in slot order into slot 0 (RS = 1: nothing to do)
template <int VEC, int RS> __device__ __forceinline__ void slot_sum(float (&acc)[VEC]) {
  if constexpr (RS > 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float tot = acc[e];
#pragma unroll
      for (int r = 1; r < RS; ++r) tot += __shfl_down(acc[e], r, 64);
      acc[e] = tot;
    }
  }
}

// RS (round 6): row slots -- RS consecutive lanes share one output chunk, lane r walks the candidate output rows oh0 + r, oh0 + r + RS, ...
// and the RS partial sums are added in slot order by slot 0 (fixed order: deterministic).  With upsampling factors 4 / 8 a thread walked 8 / 16
// rows one after the other, one memory latency each: the full-resolution decoder gradients (201 MB in) ran at 2 TB/s.
template <typename T, int RS>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ mul, T* __restrict__ dx, int B, int Hi, int Wi,
                                    int Ho, int Wo, int C, long dy_ld, long dx_ld) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int CV = C / VEC;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 >= (long)B * Hi * Wi * CV * RS) return;  // (whole groups of RS lanes: the total is a multiple of RS)
  const int rslot = (int)(i0 % RS);
  const long i = i0 / RS;
  const int cv = (int)(i % CV);
  long t = i / CV;
  const int wi = (int)(t % Wi); t /= Wi;
  const int hi = (int)(t % Hi);
  const int b = (int)(t / Hi);
  // candidate output range: src in (hi-1, hi+1)  <=>  o in ((hi-0.5)*s-0.5, (hi+1.5)*s-0.5), s = out/in  (+ clamp effects at borders)
  const float sh = (float)Ho / (float)Hi, sw = (float)Wo / (float)Wi;
  int oh0 = (int)floorf(((float)hi - 0.5f) * sh - 0.5f) - 1, oh1 = (int)ceilf(((float)hi + 1.5f) * sh - 0.5f) + 1;
  int ow0 = (int)floorf(((float)wi - 0.5f) * sw - 0.5f) - 1, ow1 = (int)ceilf(((float)wi + 1.5f) * sw - 0.5f) + 1;
  if (oh0 < 0) oh0 = 0;
  if (ow0 < 0) ow0 = 0;
  if (oh1 > Ho - 1) oh1 = Ho - 1;
  if (ow1 > Wo - 1) ow1 = Wo - 1;
  if (hi == 0) oh0 = 0;
  if (wi == 0) ow0 = 0;
  if (hi == Hi - 1) oh1 = Ho - 1;
  if (wi == Wi - 1) ow1 = Wo - 1;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  // The column weights do not depend on the output row: computed once per thread (<= MAXR candidates: upsampling factors <= 8), not once
  // per candidate pixel -- the index arithmetic was most of this kernel (8 launches of ~100 us on the chain at the start of backward).
  constexpr int MAXR = 24;
  const int nw = ow1 - ow0 + 1;
  if (nw <= MAXR) {
    float wwv[MAXR];
#pragma unroll
    for (int j = 0; j < MAXR; ++j) {
      float ww = 0.f;
      if (j < nw) {
        int w0, w1; float lw;
        bilin_coord(ow0 + j, Wi, Wo, w0, w1, lw);
        if (w0 == wi) ww += 1.f - lw;
        if (w1 == wi) ww += lw;
      }
      wwv[j] = ww;
    }
    for (int oh = oh0 + rslot; oh <= oh1; oh += RS) {
      int h0, h1; float lh;
      bilin_coord(oh, Hi, Ho, h0, h1, lh);
      float wh = 0.f;
      if (h0 == hi) wh += 1.f - lh;
      if (h1 == hi) wh += lh;
      if (wh == 0.f) continue;
      const size_t rowbase = ((size_t)b * Ho + oh) * Wo + ow0;
#pragma unroll
      for (int j = 0; j < MAXR; ++j) {
        if (j >= nw || wwv[j] == 0.f) continue;
        float g[VEC];
        load_vec<T>(dy + (rowbase + j) * dy_ld + cv * VEC, g);
        const float wt = wh * wwv[j] * (mul ? mul[(size_t)oh * Wo + ow0 + j] : 1.f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += wt * g[e];
      }
    }
    slot_sum<VEC, RS>(acc);
    if (rslot == 0) store_vec<T>(dx + (((size_t)b * Hi + hi) * Wi + wi) * dx_ld + cv * VEC, acc);
    return;
  }
  for (int oh = oh0 + rslot; oh <= oh1; oh += RS) {
    int h0, h1; float lh;
    bilin_coord(oh, Hi, Ho, h0, h1, lh);
    float wh = 0.f;
    if (h0 == hi) wh += 1.f - lh;
    if (h1 == hi) wh += lh;
    if (wh == 0.f) continue;
    for (int ow = ow0; ow <= ow1; ++ow) {
      int w0, w1; float lw;
      bilin_coord(ow, Wi, Wo, w0, w1, lw);
      float ww = 0.f;
      if (w0 == wi) ww += 1.f - lw;
      if (w1 == wi) ww += lw;
      if (ww == 0.f) continue;
      float g[VEC];
      load_vec<T>(dy + (((size_t)b * Ho + oh) * Wo + ow) * dy_ld + cv * VEC, g);
      const float wt = wh * ww * (mul ? mul[(size_t)oh * Wo + ow] : 1.f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += wt * g[e];
    }
  }
  slot_sum<VEC, RS>(acc);
  if (rslot == 0) store_vec<T>(dx + (((size_t)b * Hi + hi) * Wi + wi) * dx_ld + cv * VEC, acc);
}

extern "C" int tfpp_bilinear_bwd(const void* dy, const float* mul, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int64_t dy_ld,
                                 int64_t dx_ld, int dtype, void* stream) {
  if (!dy || !dx) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool rs4 = Ho >= 4 * Hi;  // >= 8 candidate rows per input pixel: four row slots per output chunk
  if (dtype == TFPP_F32) {
    if (C % 4 || dy_ld % 4 || dx_ld % 4) return TFPP_EINVAL;
    if (rs4) hipLaunchKernelGGL((bilinear_bwd_kernel<float, 4>), grid1d((long)B * Hi * Wi * (C / 4) * 4), dim3(PW_THREADS), 0, st, (const float*)dy, mul, (float*)dx, B, Hi, Wi, Ho, Wo, C, (long)dy_ld, (long)dx_ld);
    else hipLaunchKernelGGL((bilinear_bwd_kernel<float, 1>), grid1d((long)B * Hi * Wi * (C / 4)), dim3(PW_THREADS), 0, st, (const float*)dy, mul, (float*)dx, B, Hi, Wi, Ho, Wo, C, (long)dy_ld, (long)dx_ld);
  } else {
    if (C % 8 || dy_ld % 8 || dx_ld % 8) return TFPP_EINVAL;
    if (rs4) hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, 4>), grid1d((long)B * Hi * Wi * (C / 8) * 4), dim3(PW_THREADS), 0, st, (const bf16_t*)dy, mul, (bf16_t*)dx, B, Hi, Wi, Ho, Wo, C, (long)dy_ld, (long)dx_ld);
    else hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t, 1>), grid1d((long)B * Hi * Wi * (C / 8)), dim3(PW_THREADS), 0, st, (const bf16_t*)dy, mul, (bf16_t*)dx, B, Hi, Wi, Ho, Wo, C, (long)dy_ld, (long)dx_ld);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// squeeze-excite
// ---------------------------------------------------------------------------------------------------------------
// partial[b][blockIdx.x][c] = sum_{this workgroup's rows of image b} x[b,row,c] * (y ? y[b,row,c] : 1); column-fixed layout
// acc[e] = sum over this thread's rows of image b of x (* y): the row loop shared by hw_reduce_kernel and the fused squeeze-excite kernels
template <typename T, bool DOT>
__device__ __forceinline__ void hw_accumulate(const T* __restrict__ x, const T* __restrict__ y, int b, int HW, int CV, int rp, int rr, int cv,
                                              bool active, float (&acc)[ElemTraits<T>::VEC]) {
  constexpr int VEC = ElemTraits<T>::VEC;
  const int C = CV * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  if (active) {
    const size_t base = (size_t)b * HW * C + cv * VEC;
    const int stride = (int)gridDim.x * rp;
    constexpr int U = DOT ? 2 : 4;
    int r = (int)blockIdx.x * rp + rr;
    for (; r + (U - 1) * stride < HW; r += U * stride) {
      uint4 xv[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t off = base + (size_t)(r + u * stride) * C;
        xv[u] = *reinterpret_cast<const uint4*>(x + off);
        if (DOT) yv[u] = *reinterpret_cast<const uint4*>(y + off);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[VEC], w[VEC];
        unpack16<T>(xv[u], v);
        if (DOT) {
          unpack16<T>(yv[u], w);
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += v[e] * w[e];
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += v[e];
        }
      }
    }
    for (; r < HW; r += stride) {
      const size_t off = base + (size_t)r * C;
      float v[VEC], w[VEC];
      load_vec<T>(x + off, v);
      if (DOT) {
        load_vec<T>(y + off, w);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[e] * w[e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[e];
      }
    }
  }
}

template <typename T, bool DOT>
__global__ void hw_reduce_kernel(const T* __restrict__ x, const T* __restrict__ y, float* __restrict__ partial, int HW, int CV, int sw, int rp) {
  constexpr int VEC = ElemTraits<T>::VEC;
  int rr, cv;
  const bool active = col_thread(sw, rp, CV, rr, cv);
  const int b = blockIdx.z, C = CV * VEC;
  float acc[VEC];
  hw_accumulate<T, DOT>(x, y, b, HW, CV, rp, rr, cv, active, acc);
  __shared__ float sm[VEC * 256];
  col_block_reduce<VEC>(acc, sw, rp, rr, sm);
  if (active && rr == 0) {
    float* o = partial + ((size_t)b * gridDim.x + blockIdx.x) * C + cv * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = acc[e];
  }
}

// (Round 5 measured a fused squeeze-excite -- squeeze + fc1 + fc2 in ONE launch, the workgroup that draws the last ticket of a sample runs the two
// tiny fully-connected layers -- and removed it again: alone on the chip it wins only for stage 1 (20 vs 29 us), stage 3 takes 66 us and stage 4
// 231 us against 30 us for the four launches below (a chain of dependent load round trips on ONE CU, profiles/r05_se_micro.txt); in the step
// even the stage-1-only variant cost +0.7 ms and the bs = 1 forward did not move (profiles/r05_ab_se_*.txt).)
// hw_reduce + col_final in ONE launch (round 5): the (<= 16) row-block workgroups of a sample publish their partial sums and draw a ticket per
// sample; the workgroup that draws the last one adds the partials in row-block order -- all of them fetched in one round trip
// (grid_fetch_sum16) -- scales and writes out[b][c].  Fixed order: bit-reproducible.  Saves the second launch (5 us + the gap in front of it) of
// every squeeze (forward) and every dgate reduction (backward) of the 42 squeeze-excite blocks.
template <typename T, bool DOT>
__global__ __launch_bounds__(256) void hw_reduce_ticket_kernel(const T* __restrict__ x, const T* __restrict__ y, float* __restrict__ partial,
                                                               float* __restrict__ out, unsigned* __restrict__ tickets, int HW, int CV, int sw, int rp,
                                                               float mul) {
  constexpr int VEC = ElemTraits<T>::VEC;
  int rr, cv;
  const bool active = col_thread(sw, rp, CV, rr, cv);
  const int b = blockIdx.z, C = CV * VEC;
  float acc[VEC];
  hw_accumulate<T, DOT>(x, y, b, HW, CV, rp, rr, cv, active, acc);
  __shared__ float sm[VEC * 256];
  col_block_reduce<VEC>(acc, sw, rp, rr, sm);
  if (active && rr == 0) {
    float* o = partial + ((size_t)b * gridDim.x + blockIdx.x) * C + cv * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) grid_publish(o + e, acc[e]);
  }
  if (!grid_last_ticket(tickets + b, gridDim.x * gridDim.y)) return;
  for (int c = threadIdx.x; c < C; c += 256)
    out[(size_t)b * C + c] = grid_fetch_sum16(partial + (size_t)b * gridDim.x * C + c, C, (int)gridDim.x) * mul;
}

template <typename T>
static int launch_hw_reduce_ticket(const void* x, const void* y, float* out, float* scratch, float* tickets, int B, int HW, int C, float mulv, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC || B < 1 || B > TFPP_GRIDSUM_TICKETS) return TFPP_EINVAL;
  const ColLayout l = col_layout(C / VEC);
  int nb = col_blocks_x(HW, l, y ? 4 : 8, 4096, B);
  if (nb > 16) nb = 16;  // one batch of agent-scope loads per channel in the tail
  dim3 grid((unsigned)nb, (unsigned)l.ny, (unsigned)B);
  if (y) hipLaunchKernelGGL((hw_reduce_ticket_kernel<T, true>), grid, dim3(256), 0, st, (const T*)x, (const T*)y, scratch, out, reinterpret_cast<unsigned*>(tickets), HW, C / VEC, l.sw, l.rp, mulv);
  else hipLaunchKernelGGL((hw_reduce_ticket_kernel<T, false>), grid, dim3(256), 0, st, (const T*)x, (const T*)y, scratch, out, reinterpret_cast<unsigned*>(tickets), HW, C / VEC, l.sw, l.rp, mulv);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// one-launch forms of tfpp_mean_hw / tfpp_se_dgate: ticket_scratch = a tfpp_gridsum_scratch_floats() buffer (zero before the first use, left at
// zero, one per stream); B <= 64, else TFPP_EINVAL (callers use the two-launch forms)
extern "C" int tfpp_mean_hw_ticket(const void* x, float* out, float* scratch, float* ticket_scratch, int B, int HW, int C, int dtype, void* stream) {
  if (!x || !out || !scratch || !ticket_scratch) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_hw_reduce_ticket<float>(x, nullptr, out, scratch, ticket_scratch, B, HW, C, 1.f / (float)HW, st)
                           : launch_hw_reduce_ticket<bf16_t>(x, nullptr, out, scratch, ticket_scratch, B, HW, C, 1.f / (float)HW, st);
}
extern "C" int tfpp_se_dgate_ticket(const void* dy, const void* x, float* dgate, float* scratch, float* ticket_scratch, int B, int HW, int C, int dtype,
                                    void* stream) {
  if (!dy || !x || !dgate || !scratch || !ticket_scratch) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_hw_reduce_ticket<float>(dy, x, dgate, scratch, ticket_scratch, B, HW, C, 1.f, st)
                           : launch_hw_reduce_ticket<bf16_t>(dy, x, dgate, scratch, ticket_scratch, B, HW, C, 1.f, st);
}

template <typename T>
static int launch_hw_reduce(const void* x, const void* y, float* out, float* scratch, int B, int HW, int C, float mulv, hipStream_t st) {
  constexpr int VEC = ElemTraits<T>::VEC;
  if (C % VEC) return TFPP_EINVAL;
  const ColLayout l = col_layout(C / VEC);
  int nb = col_blocks_x(HW, l, y ? 4 : 8, 4096, B);
  if (nb > HW_MAX_PARTIALS) nb = HW_MAX_PARTIALS;
  dim3 grid((unsigned)nb, (unsigned)l.ny, (unsigned)B);
  if (y) hipLaunchKernelGGL((hw_reduce_kernel<T, true>), grid, dim3(256), 0, st, (const T*)x, (const T*)y, scratch, HW, C / VEC, l.sw, l.rp);
  else hipLaunchKernelGGL((hw_reduce_kernel<T, false>), grid, dim3(256), 0, st, (const T*)x, (const T*)y, scratch, HW, C / VEC, l.sw, l.rp);
  hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0, st, scratch, out, nb, C, mulv, 0);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_mean_hw(const void* x, float* out, float* scratch, int B, int HW, int C, int dtype, void* stream) {
  if (!x || !out || !scratch) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_hw_reduce<float>(x, nullptr, out, scratch, B, HW, C, 1.f / (float)HW, st)
                           : launch_hw_reduce<bf16_t>(x, nullptr, out, scratch, B, HW, C, 1.f / (float)HW, st);
}

extern "C" int tfpp_se_dgate(const void* dy, const void* x, float* dgate, float* scratch, int B, int HW, int C, int dtype, void* stream) {
  if (!dy || !x || !dgate || !scratch) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return dtype == TFPP_F32 ? launch_hw_reduce<float>(dy, x, dgate, scratch, B, HW, C, 1.f, st)
                           : launch_hw_reduce<bf16_t>(dy, x, dgate, scratch, B, HW, C, 1.f, st);
}

// squeeze-excite gate: hidden = relu(W1 pool + b1) [one wave per (b,j)] ; gate = sigmoid(W2 hidden + b2) [thread per (b,c)]
__global__ void se_hidden_kernel(const float* __restrict__ pool, const float* __restrict__ w1, const float* __restrict__ b1,
                                 float* __restrict__ hidden, int B, int C, int RD) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= B * RD) return;
  const int b = w / RD, j = w - b * RD;
  const float* wr = w1 + (size_t)j * C;
  const float* pr = pool + (size_t)b * C;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains: the loads of four 64-channel segments are in flight together
  int c = lane;
  for (; c + 192 < C; c += 256) {
    s0 += wr[c] * pr[c];
    s1 += wr[c + 64] * pr[c + 64];
    s2 += wr[c + 128] * pr[c + 128];
    s3 += wr[c + 192] * pr[c + 192];
  }
  for (; c < C; c += 64) s0 += wr[c] * pr[c];
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) {
    s += b1[j];
    hidden[w] = s > 0.f ? s : 0.f;
  }
}

// gate[b,c] = sigmoid(b2[c] + sum_j W2[c][j] hidden[b][j]): one WAVE per (b, c), lanes over j -- the rows of W2 are RD contiguous floats, read as
// whole 256-byte segments.  (Rounds 1-4 ran one THREAD per channel: every lane walked its own row, RD strided loads of one cache line each --
// 8 us for a 576 x 144 matrix-vector product at bs = 1.  A/B in one call, profiles/r05_ab_se_gate.txt: bs = 1 forward 3.95 -> 3.87 ms bf16,
// 5.95 -> 5.86 ms fp32, training step unchanged.)
__global__ void se_gate_kernel(const float* __restrict__ hidden, const float* __restrict__ w2, const float* __restrict__ b2,
                               float* __restrict__ gate, int C, int RD) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y, c = (int)blockIdx.x * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
  if (c >= C) return;
  const float* wr = w2 + (size_t)c * RD;
  const float* h = hidden + (size_t)b * RD;
  float s = 0.f;
  for (int j = lane; j < RD; j += 64) s += wr[j] * h[j];
  s = wave_sum(s);
  if (lane == 0) gate[(size_t)b * C + c] = 1.f / (1.f + __expf(-(s + b2[c])));
}

extern "C" int tfpp_se_gate_fwd(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, float* hidden,
                                float* gate, int B, int C, int RD, void* stream) {
  if (!pool || !w1 || !w2 || !hidden || !gate) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_hidden_kernel, dim3((B * RD + 3) / 4), dim3(256), 0, st, pool, w1, b1, hidden, B, C, RD);
  hipLaunchKernelGGL(se_gate_kernel, dim3((C + 3) / 4, B), dim3(256), 0, st, hidden, w2, b2, gate, C, RD);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// backward of the gate MLP.  gd[b,c] = dgate*g*(1-g).  dz1[b,j] = (hidden>0) * sum_c gd[b,c] w2[c,j].
// One 1024-thread workgroup per (sample, 64 hidden units): lane jl reads W2[c][j0 + jl] -- 64 consecutive floats of row c, one 256-byte segment --
// for the channels c = cg, cg + 16, ... of its group (16 groups of 64 lanes), twelve loads in flight; the groups are summed through LDS in a fixed
// order.  (Rounds 1-4: one wave per (b, j) with the lanes over c, 64 different cache lines per load.  Round 5: this layout with 4 groups and 4
// loads in flight -- a chain of 36 dependent L2 round trips for C = 576, 17.8 us per launch in the captured bs = 12 step; 16 groups x 12 in
// flight make it 3.)
#define SE_DZ1_GROUPS 16
__global__ __launch_bounds__(64 * SE_DZ1_GROUPS) void se_dz1_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                                    const float* __restrict__ hidden, const float* __restrict__ w2,
                                                                    float* __restrict__ dz1, int B, int C, int RD, int premul) {
  constexpr int NG = SE_DZ1_GROUPS, U = 12;
  const int jl = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const int b = blockIdx.y, j = (int)blockIdx.x * 64 + jl;
  const bool ok = j < RD;
  const float* dg = dgate + (size_t)b * C;
  const float* g = gate + (size_t)b * C;
  const float* wc = w2 + (ok ? j : 0);
  float s[U];
#pragma unroll
  for (int u = 0; u < U; ++u) s[u] = 0.f;
  int c = cg;
  for (; c + (U - 1) * NG < C; c += U * NG) {
    float wv[U], gv[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { wv[u] = wc[(size_t)(c + u * NG) * RD]; gv[u] = g[c + u * NG]; dv[u] = dg[c + u * NG]; }
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] += dv[u] * (premul ? 1.f : gv[u]) * (1.f - gv[u]) * wv[u];
  }
  for (; c < C; c += NG) {
    const float g0 = g[c];
    s[0] += dg[c] * (premul ? 1.f : g0) * (1.f - g0) * wc[(size_t)c * RD];
  }
  float t = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) t += s[u];
  __shared__ float sm[NG][64];
  sm[cg][jl] = t;
  __syncthreads();
  if (cg == 0 && ok) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < NG; ++q) tot += sm[q][jl];
    dz1[(size_t)b * RD + j] = hidden[(size_t)b * RD + j] > 0.f ? tot : 0.f;
  }
}

// Single writer per output (no atomics).  Workgroups [0, nblk_dw): one thread per dw2[c][j] (+db2) / dw1[j][c] (+db1) element.
// Workgroups after that: dpool[b][c] = sum_j dz1[b,j] w1[j,c], 16 channels x 16 j-slots per workgroup (the RD-long
// reduction is the long pole of this kernel when a single thread walks it).
__global__ void se_param_grads_kernel(const float* __restrict__ dgate, const float* __restrict__ gate, const float* __restrict__ hidden,
                                      const float* __restrict__ pool, const float* __restrict__ w1, const float* __restrict__ dz1,
                                      float* __restrict__ dpool, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                      float* __restrict__ db2, int B, int C, int RD, int nblk_dw, int premul) {
  if ((int)blockIdx.x >= nblk_dw) {
    const int ctiles = (C + 15) / 16;
    const int t = blockIdx.x - nblk_dw, b = t / ctiles, c = (t - b * ctiles) * 16 + (threadIdx.x & 15), js = threadIdx.x >> 4;
    float s = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int j = js; j < RD; j += 16) s += dz1[b * RD + j] * w1[(size_t)j * C + c];
    }
    __shared__ float sm[16][17];
    sm[js][threadIdx.x & 15] = s;
    __syncthreads();
    if (js == 0 && c < C) {
      float tot = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) tot += sm[k][threadIdx.x];
      dpool[(size_t)b * C + c] = tot;
    }
    return;
  }
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned n1 = (unsigned)C * RD;
  if (i < n1) {  // dw2[c][j]
    const unsigned c = i / RD, j = i - c * RD;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float g = gate[(size_t)b * C + c];
      const float gd = dgate[(size_t)b * C + c] * (premul ? 1.f : g) * (1.f - g);
      s += gd * hidden[(size_t)b * RD + j];
      sb += gd;
    }
    dw2[i] += s;
    if (j == 0) db2[c] += sb;
  } else if (i < 2 * n1) {  // dw1[j][c]
    const unsigned k = i - n1;
    const unsigned j = k / C, c = k - j * C;
    float s = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
      const float d = dz1[(size_t)b * RD + j];
      s += d * pool[(size_t)b * C + c];
      sb += d;
    }
    dw1[k] += s;
    if (c == 0) db1[j] += sb;
  }
}

static int se_gate_bwd_impl(const float* dgate, const float* gate, const float* hidden, const float* pool, const float* w1,
                           const float* w2, float* dz1_scratch, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int B,
                           int C, int RD, int premul, void* stream) {
  if (!dgate || !gate || !hidden || !pool || !dpool || !dz1_scratch) return TFPP_EINVAL;
  if (2l * C * RD >= (1l << 31)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(se_dz1_kernel, dim3((unsigned)((RD + 63) / 64), (unsigned)B), dim3(64 * SE_DZ1_GROUPS), 0, st, dgate, gate, hidden, w2, dz1_scratch, B, C, RD, premul);
  const int nblk_dw = (int)((2l * C * RD + 255) / 256);
  hipLaunchKernelGGL(se_param_grads_kernel, dim3((unsigned)(nblk_dw + B * ((C + 15) / 16))), dim3(256), 0, st, dgate, gate, hidden, pool, w1,
                     dz1_scratch, dpool, dw1, db1, dw2, db2, B, C, RD, nblk_dw, premul);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_se_gate_bwd(const float* dgate, const float* gate, const float* hidden, const float* pool, const float* w1,
                                const float* w2, float* dz1_scratch, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int B,
                                int C, int RD, void* stream) {
  return se_gate_bwd_impl(dgate, gate, hidden, pool, w1, w2, dz1_scratch, dpool, dw1, db1, dw2, db2, B, C, RD, 0, stream);
}
// the same with dgate_g = dgate * gate handed in (= sum_hw dy * y for the GATED tensor y the forward pass wrote: tfpp.h)
extern "C" int tfpp_se_gate_bwd_premul(const float* dgate_g, const float* gate, const float* hidden, const float* pool, const float* w1,
                                       const float* w2, float* dz1_scratch, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int B,
                                       int C, int RD, void* stream) {
  return se_gate_bwd_impl(dgate_g, gate, hidden, pool, w1, w2, dz1_scratch, dpool, dw1, db1, dw2, db2, B, C, RD, 1, stream);
}

// dx = dy * gate[b,c] + dpool[b,c] / HW
template <typename T>
__global__ void se_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool,
                                    T* __restrict__ dx, long nvec, int CV, long HW) {
  constexpr int VEC = ElemTraits<T>::VEC;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const float inv = 1.f / (float)HW;
  for (; i < nvec; i += stride) {
    const long row = i / CV;
    const int c0 = (int)(i - row * CV) * VEC;
    const long b = row / HW;
    float v[VEC];
    load_vec<T>(dy + i * VEC, v);
    const float* g = gate + b * (long)CV * VEC + c0;
    const float* dp = dpool + b * (long)CV * VEC + c0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = v[e] * g[e] + dp[e] * inv;
    store_vec<T>(dx + i * VEC, v);
  }
}

extern "C" int tfpp_se_bwd_apply(const void* dy, const float* gate, const float* dpool, void* dx, int B, int HW, int C, int dtype, void* stream) {
  if (!dy || !gate || !dpool || !dx) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TFPP_F32) {
    if (C % 4) return TFPP_EINVAL;
    const long nvec = (long)B * HW * (C / 4);
    hipLaunchKernelGGL(se_bwd_apply_kernel<float>, grid_stride(nvec), dim3(PW_THREADS), 0, st, (const float*)dy, gate, dpool, (float*)dx, nvec, C / 4, (long)HW);
  } else {
    if (C % 8) return TFPP_EINVAL;
    const long nvec = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(se_bwd_apply_kernel<bf16_t>, grid_stride(nvec), dim3(PW_THREADS), 0, st, (const bf16_t*)dy, gate, dpool, (bf16_t*)dx, nvec, C / 8, (long)HW);
  }
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// small token plumbing: dst[b*dst_bs + dst_off + i] (+)= src[b*src_bs + src_off + i], i in [0, n)  (concat / split /
// broadcast of token blocks; src_bs = 0 broadcasts a learned query over the batch)
// ---------------------------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ void copy_rows_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int B, long n, long src_bs, long src_off, long dst_bs,
                                 long dst_off, int accumulate) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * n, stride = (long)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const long b = i / n, k = i - b * n;
    float v = ElemTraits<TI>::to_f(src[b * src_bs + src_off + k]);
    TO* d = dst + b * dst_bs + dst_off + k;
    if (accumulate) v += ElemTraits<TO>::to_f(*d);
    *d = ElemTraits<TO>::from_f(v);
  }
}

extern "C" int tfpp_copy_rows(const void* src, void* dst, int B, int64_t n, int64_t src_bs, int64_t src_off, int64_t dst_bs,
                              int64_t dst_off, int accumulate, int dtype_in, int dtype_out, void* stream) {
  if (!src || !dst) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  dim3 g = grid_stride((long)B * n), b(PW_THREADS);
#define CR(TI, TO) hipLaunchKernelGGL((copy_rows_kernel<TI, TO>), g, b, 0, st, (const TI*)src, (TO*)dst, B, (long)n, (long)src_bs, (long)src_off, (long)dst_bs, (long)dst_off, accumulate)
  if (dtype_in == TFPP_F32 && dtype_out == TFPP_F32) CR(float, float);
  else if (dtype_in == TFPP_F32 && dtype_out == TFPP_BF16) CR(float, bf16_t);
  else if (dtype_in == TFPP_BF16 && dtype_out == TFPP_F32) CR(bf16_t, float);
  else if (dtype_in == TFPP_BF16 && dtype_out == TFPP_BF16) CR(bf16_t, bf16_t);
  else return TFPP_EINVAL;
#undef CR
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_zero(void* p, int64_t bytes, void* stream) {
  if (!p) return TFPP_EINVAL;
  if (bytes < 0) return TFPP_EINVAL;
  return tfpp_fill_async(p, 0, (size_t)bytes, (hipStream_t)stream);
}

extern "C" int tfpp_fill_bytes(void* p, int value, int64_t bytes, void* stream) {
  if (!p || bytes < 0) return TFPP_EINVAL;
  return tfpp_fill_async(p, value, (size_t)bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// multi-tensor weight packing: every per-step weight image (conv forward / data-gradient layouts, head-padded QKV,
// transposes, fp32 -> bf16 casts) in ONE launch, driven by a device-resident descriptor table.
// ---------------------------------------------------------------------------------------------------------------
#define PACK_ELEMS_PER_BLOCK 2048
template <typename T>
__device__ __forceinline__ void pack_one(const tfpp_pack_desc& d, long i) {
  float v = 0.f;
  if (d.kind == 0 || d.kind == 1) {  // conv weight, a = {Cout, cin_g, R, S, G, ks_pad, n_pad}
    const int Cout = d.a[0], cin_g = d.a[1], RS = d.a[2] * d.a[3], G = d.a[4], ks_pad = d.a[5], n_pad = d.a[6];
    const int n_g = Cout / G;
    if (d.kind == 0) {
      const int K = RS * ks_pad;
      const int k = (int)(i % K);
      const long t = i / K;
      const int n = (int)(t % n_pad), g = (int)(t / n_pad);
      const int rs = k / ks_pad, c = k - rs * ks_pad;
      if (n < n_g && c < cin_g) v = d.src[((size_t)(g * n_g + n) * cin_g + c) * RS + rs];
    } else {
      const int K = RS * n_pad;
      const int k = (int)(i % K);
      const long t = i / K;
      const int c = (int)(t % cin_g), g = (int)(t / cin_g);
      const int rs = k / n_pad, n = k - rs * n_pad;
      if (n < n_g) v = d.src[((size_t)(g * n_g + n) * cin_g + c) * RS + rs];
    }
    reinterpret_cast<T*>(d.dst)[i] = ElemTraits<T>::from_f(v);
  } else {  // pack2d, a = {rows_out, cols_out, transpose_in}
    const int cols_out = d.a[1];
    const int r = (int)(i / cols_out), c = (int)(i - (long)r * cols_out);
    const int sr = d.row_map ? d.row_map[r] : r, sc = d.col_map ? d.col_map[c] : c;
    if (sr >= 0 && sc >= 0) v = d.a[2] ? d.src[(size_t)sc * d.in_ld + sr] : d.src[(size_t)sr * d.in_ld + sc];
    reinterpret_cast<T*>(d.dst)[(size_t)r * d.out_ld + c] = ElemTraits<T>::from_f(v);
  }
}

// Fast paths (mode in a[7], set by tfpp_pack_desc_plan):
//   1  contiguous cast copy (1x1 forward image without padding): 8 elements per thread, 16-byte stores
//   2  tiled transpose (1x1 data-gradient image, transposed pack2d incl. head-padding maps): a workgroup owns a
//      64 (destination rows = source-contiguous index) x 32 (destination columns) tile, reads the source coalesced, turns the
//      tile in LDS and writes 16-byte row segments.  The element-wise path read such sources with a stride of a whole row.
#define PACK_TILE_R 64
#define PACK_TILE_C 32
struct PackTileGeom { int rows, cols, groups; };  // destination matrix per group
__host__ __device__ static inline PackTileGeom pack_tile_geom(const tfpp_pack_desc& d) {
  PackTileGeom g;
  if (d.kind == 1) { g.rows = d.a[1]; g.cols = d.a[6]; g.groups = d.a[4]; }   // [cin_g][n_pad] per group
  else { g.rows = d.a[0]; g.cols = d.a[1]; g.groups = 1; }                    // pack2d [rows_out][cols_out]
  return g;
}

template <typename T> __device__ __forceinline__ void pack_store8(T* dst, const float* v, int nvalid);
template <> __device__ __forceinline__ void pack_store8<bf16_t>(bf16_t* dst, const float* v, int nvalid) {
  if (nvalid >= 8 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) store_vec<bf16_t>(dst, v);
  else for (int e = 0; e < nvalid && e < 8; ++e) dst[e] = f2bf(v[e]);
}
template <> __device__ __forceinline__ void pack_store8<float>(float* dst, const float* v, int nvalid) {
  if (nvalid >= 8 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) { store_vec<float>(dst, v); store_vec<float>(dst + 4, v + 4); }
  else for (int e = 0; e < nvalid && e < 8; ++e) dst[e] = v[e];
}

template <typename T> __device__ __forceinline__ void pack_tile_transpose(const tfpp_pack_desc& d, long rel_block, float* lds) {
  const PackTileGeom gm = pack_tile_geom(d);
  const int tiles_c = (gm.cols + PACK_TILE_C - 1) / PACK_TILE_C, tiles_r = (gm.rows + PACK_TILE_R - 1) / PACK_TILE_R;
  const int tc = (int)(rel_block % tiles_c);
  const long t2 = rel_block / tiles_c;
  const int tr = (int)(t2 % tiles_r), g = (int)(t2 / tiles_r);
  const int r0 = tr * PACK_TILE_R, c0 = tc * PACK_TILE_C, tid = threadIdx.x;
  {  // load: thread = (destination column, 8 consecutive destination rows = 8 consecutive source elements)
    const int col = tid >> 3, rch = tid & 7, c = c0 + col;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (c < gm.cols) {
      if (d.kind == 1) {
        const int n_g = d.a[0] / d.a[4], cin_g = d.a[1];
        if (c < n_g) {
          const float* sp = d.src + ((size_t)g * n_g + c) * cin_g;
#pragma unroll
          for (int e = 0; e < 8; ++e) { const int r = r0 + rch * 8 + e; if (r < gm.rows) v[e] = sp[r]; }
        }
      } else {
        const int sc = d.col_map ? d.col_map[c] : c;
        if (sc >= 0) {
          const float* sp = d.src + (size_t)sc * d.in_ld;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = r0 + rch * 8 + e;
            if (r < gm.rows) { const int sr = d.row_map ? d.row_map[r] : r; if (sr >= 0) v[e] = sp[sr]; }
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) lds[(rch * 8 + e) * (PACK_TILE_C + 1) + col] = v[e];
  }
  __syncthreads();
  {  // store: thread = (destination row, 8 consecutive destination columns)
    const int row = tid >> 2, cch = tid & 3, r = r0 + row, c = c0 + cch * 8;
    if (r < gm.rows && c < gm.cols) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = lds[row * (PACK_TILE_C + 1) + cch * 8 + e];
      const size_t off = d.kind == 1 ? ((size_t)g * gm.rows + r) * gm.cols + c : (size_t)r * d.out_ld + c;
      pack_store8<T>(reinterpret_cast<T*>(d.dst) + off, v, gm.cols - c);
    }
  }
}

__global__ void pack_multi_kernel(const tfpp_pack_desc* __restrict__ descs, int n) {
  // binary search: last descriptor with blk_start <= blockIdx.x
  int lo = 0, hi = n - 1;
  const long b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk_start <= b) lo = mid;
    else hi = mid - 1;
  }
  const tfpp_pack_desc d = descs[lo];
  const int mode = d.a[7];
  if (mode == 2) {
    __shared__ float lds[PACK_TILE_R * (PACK_TILE_C + 1)];
    if (d.dtype == TFPP_F32) pack_tile_transpose<float>(d, b - d.blk_start, lds);
    else pack_tile_transpose<bf16_t>(d, b - d.blk_start, lds);
    return;
  }
  const long base = (b - d.blk_start) * PACK_ELEMS_PER_BLOCK;
  if (mode == 1) {
    const long i = base + (long)threadIdx.x * 8;
    if (i < d.total) {
      float v[8];
      const int nv = d.total - i < 8 ? (int)(d.total - i) : 8;
      for (int e = 0; e < 8; ++e) v[e] = e < nv ? d.src[i + e] : 0.f;
      if (d.dtype == TFPP_F32) pack_store8<float>(reinterpret_cast<float*>(d.dst) + i, v, nv);
      else pack_store8<bf16_t>(reinterpret_cast<bf16_t*>(d.dst) + i, v, nv);
    }
    return;
  }
  for (int e = threadIdx.x; e < PACK_ELEMS_PER_BLOCK; e += blockDim.x) {
    const long i = base + e;
    if (i >= d.total) break;
    if (d.dtype == TFPP_F32) pack_one<float>(d, i);
    else pack_one<bf16_t>(d, i);
  }
}

// Host side of the table: choose the path for a descriptor (written to a[7]) and return the workgroups it needs.
extern "C" int tfpp_pack_desc_plan(tfpp_pack_desc* d) {
  if (!d) return TFPP_EINVAL;
  int mode = 0;
  if (d->kind == 0 && d->a[2] * d->a[3] == 1 && d->a[5] == d->a[1] && d->a[6] * d->a[4] == d->a[0]) mode = 1;
  else if (d->kind == 1 && d->a[2] * d->a[3] == 1) mode = 2;
  else if (d->kind == 2 && d->a[2] != 0) mode = 2;
  d->a[7] = mode;
  if (mode == 2) {
    const PackTileGeom g = pack_tile_geom(*d);
    return g.groups * ((g.rows + PACK_TILE_R - 1) / PACK_TILE_R) * ((g.cols + PACK_TILE_C - 1) / PACK_TILE_C);
  }
  return (int)((d->total + PACK_ELEMS_PER_BLOCK - 1) / PACK_ELEMS_PER_BLOCK);
}

extern "C" int tfpp_pack_elems_per_block(void) { return PACK_ELEMS_PER_BLOCK; }

extern "C" int tfpp_pack_multi(const tfpp_pack_desc* descs_dev, int n, int64_t total_blocks, void* stream) {
  if (!descs_dev || n < 1 || total_blocks < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
  TFPP_CHECK_LAUNCH();
  return 0;
}
