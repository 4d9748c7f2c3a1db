// Kernels of the fp32 planning head (team_code/model.py:137-146,333-358: nn.TransformerDecoder over 11 / 8 queries and 65 memory tokens).
// The head is 0.1 % of the step's FLOPs and, launch by launch, 10 % of its time: every kernel below replaces a run of dependent launches
// of the general kernels (batched GEMM -> softmax -> batched GEMM ...) by one.
#include "common.h"
#include "../../include/tfpp.h"

// ---------------------------------------------------------------------------------------------------------------
// Attention core of nn.MultiheadAttention for a handful of tokens: tq <= 16 queries, tk <= 96 keys, head dim <= 32, fp32.
// One workgroup per (sample, head); q, k, v, scores and probabilities live in LDS (K / V rows padded to an odd stride: conflict-free
// whether the threads of a wave walk keys or channels).  q / k / v are head-major column slices of token matrices, element (b, t, h, e) at
// base + (b*T + t) * ld + h*d + e, exactly as tfpp_bgemm was pointed at them.  Dropout: hash, seed and element index (row * tk + key,
// row = (b*nh + h)*tq + query) of tfpp_softmax_fwd, so this path and the three-launch path draw the same masks.
// forward saves the probabilities BEFORE dropout, p_save[B*nh*tq][tk] (3 KB per head), for the backward.
// ---------------------------------------------------------------------------------------------------------------
#define SA_MAXQ 16
#define SA_MAXK 96
#define SA_MAXD 32

__global__ void __launch_bounds__(256)
small_attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, float* __restrict__ o,
                      float* __restrict__ p_save, int nh, int tq, int tk, int d, long ld_q, long ld_kv, long ld_o, float scale, float p_drop,
                      float inv_keep, unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  __shared__ float Q[SA_MAXQ * SA_MAXD], K[SA_MAXK * (SA_MAXD + 1)], V[SA_MAXK * SA_MAXD], S[SA_MAXQ * (SA_MAXK + 1)];
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;  // per-step device counter: hipGraph replays draw fresh masks
  const int bh = blockIdx.x, b = bh / nh, h = bh % nh, tid = threadIdx.x;
  const int ldk = d + 1, lds = tk + 1;
  for (int e = tid; e < tq * d; e += 256) {
    const int i = e / d, c = e % d;
    Q[i * d + c] = q[((long)b * tq + i) * ld_q + h * d + c];
  }
  for (int e = tid; e < tk * d; e += 256) {
    const int j = e / d, c = e % d;
    const long off = ((long)b * tk + j) * ld_kv + h * d + c;
    K[j * ldk + c] = k[off];
    V[j * d + c] = v[off];
  }
  __syncthreads();
  for (int e = tid; e < tq * tk; e += 256) {
    const int i = e / tk, j = e % tk;
    float s = 0.f;
    for (int c = 0; c < d; ++c) s += Q[i * d + c] * K[j * ldk + c];
    S[i * lds + j] = s * scale;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int i = wave; i < tq; i += 4) {  // softmax: one wave per query row, two keys per lane
    const int j0 = lane, j1 = lane + 64;
    const float v0 = j0 < tk ? S[i * lds + j0] : -3.0e38f, v1 = j1 < tk ? S[i * lds + j1] : -3.0e38f;
    const float mx = wave_max(fmaxf(v0, v1));
    const float e0 = j0 < tk ? __expf(v0 - mx) : 0.f, e1 = j1 < tk ? __expf(v1 - mx) : 0.f;
    const float inv = 1.f / wave_sum(e0 + e1);
    const unsigned long long row = (unsigned long long)bh * tq + i;
    if (j0 < tk) {
      const float p = e0 * inv;
      p_save[row * tk + j0] = p;
      S[i * lds + j0] = p_drop > 0.f ? p * dropout_scale(seed, row * tk + j0, p_drop, inv_keep) : p;
    }
    if (j1 < tk) {
      const float p = e1 * inv;
      p_save[row * tk + j1] = p;
      S[i * lds + j1] = p_drop > 0.f ? p * dropout_scale(seed, row * tk + j1, p_drop, inv_keep) : p;
    }
  }
  __syncthreads();
  for (int e = tid; e < tq * d; e += 256) {
    const int i = e / d, c = e % d;
    float s = 0.f;
    for (int j = 0; j < tk; ++j) s += S[i * lds + j] * V[j * d + c];
    o[((long)b * tq + i) * ld_o + h * d + c] = s;
  }
}

// backward: dV = Pd^T dO;  dPd = dO V^T;  dP = dPd * mask;  dS = scale * P .* (dP - sum_j dP_j P_j);  dQ = dS K;  dK = dS^T Q
__global__ void __launch_bounds__(256)
small_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, const float* __restrict__ p_save,
                      const float* __restrict__ d_o, float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv, int nh, int tq,
                      int tk, int d, long ld_q, long ld_kv, long ld_o, float scale, float p_drop, float inv_keep, unsigned long long seed,
                      const unsigned long long* __restrict__ seed_off) {
  __shared__ float Q[SA_MAXQ * SA_MAXD], K[SA_MAXK * (SA_MAXD + 1)], V[SA_MAXK * (SA_MAXD + 1)], G[SA_MAXQ * SA_MAXD];
  __shared__ float P[SA_MAXQ * (SA_MAXK + 1)], PD[SA_MAXQ * (SA_MAXK + 1)], DS[SA_MAXQ * (SA_MAXK + 1)];
  if (seed_off) seed += *seed_off * 0x9E3779B97F4A7C15ull;
  const int bh = blockIdx.x, b = bh / nh, h = bh % nh, tid = threadIdx.x;
  const int ldk = d + 1, lds = tk + 1;
  for (int e = tid; e < tq * d; e += 256) {
    const int i = e / d, c = e % d;
    Q[i * d + c] = q[((long)b * tq + i) * ld_q + h * d + c];
    G[i * d + c] = d_o[((long)b * tq + i) * ld_o + h * d + c];
  }
  for (int e = tid; e < tk * d; e += 256) {
    const int j = e / d, c = e % d;
    const long off = ((long)b * tk + j) * ld_kv + h * d + c;
    K[j * ldk + c] = k[off];
    V[j * ldk + c] = v[off];
  }
  for (int e = tid; e < tq * tk; e += 256) {
    const int i = e / tk, j = e % tk;
    P[i * lds + j] = p_save[((long)bh * tq + i) * tk + j];
  }
  __syncthreads();
  for (int e = tid; e < tq * tk; e += 256) {
    const int i = e / tk, j = e % tk;
    float s = 0.f;
    for (int c = 0; c < d; ++c) s += G[i * d + c] * V[j * ldk + c];
    const float m = p_drop > 0.f ? dropout_scale(seed, ((unsigned long long)bh * tq + i) * tk + j, p_drop, inv_keep) : 1.f;
    PD[i * lds + j] = P[i * lds + j] * m;
    DS[i * lds + j] = s * m;  // dP
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int i = wave; i < tq; i += 4) {
    const int j0 = lane, j1 = lane + 64;
    const float p0 = j0 < tk ? P[i * lds + j0] : 0.f, p1 = j1 < tk ? P[i * lds + j1] : 0.f;
    const float g0 = j0 < tk ? DS[i * lds + j0] : 0.f, g1 = j1 < tk ? DS[i * lds + j1] : 0.f;
    const float dot = wave_sum(g0 * p0 + g1 * p1);
    if (j0 < tk) DS[i * lds + j0] = scale * p0 * (g0 - dot);
    if (j1 < tk) DS[i * lds + j1] = scale * p1 * (g1 - dot);
  }
  __syncthreads();
  for (int e = tid; e < tk * d; e += 256) {
    const int j = e / d, c = e % d;
    float sv = 0.f, sk = 0.f;
    for (int i = 0; i < tq; ++i) {
      sv += PD[i * lds + j] * G[i * d + c];
      sk += DS[i * lds + j] * Q[i * d + c];
    }
    const long off = ((long)b * tk + j) * ld_kv + h * d + c;
    dv[off] = sv;
    dk[off] = sk;
  }
  for (int e = tid; e < tq * d; e += 256) {
    const int i = e / d, c = e % d;
    float s = 0.f;
    for (int j = 0; j < tk; ++j) s += DS[i * lds + j] * K[j * ldk + c];
    dq[((long)b * tq + i) * ld_q + h * d + c] = s;
  }
}

extern "C" int tfpp_small_attn_supported(int tq, int tk, int d) {
  return tq >= 1 && tk >= 1 && d >= 1 && tq <= SA_MAXQ && tk <= SA_MAXK && d <= SA_MAXD;
}

extern "C" int tfpp_small_attn_fwd(const float* q, const float* k, const float* v, float* o, float* p_save, int B, int nh, int tq, int tk, int d,
                                   int64_t ld_q, int64_t ld_kv, int64_t ld_o, float scale, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                                   void* stream) {
  if (!q || !k || !v || !o || !p_save || B < 1 || nh < 1 || !tfpp_small_attn_supported(tq, tk, d) || p_drop < 0.f || p_drop >= 1.f) return TFPP_EINVAL;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipLaunchKernelGGL(small_attn_fwd_kernel, dim3(B * nh), dim3(256), 0, (hipStream_t)stream, q, k, v, o, p_save, nh, tq, tk, d, (long)ld_q, (long)ld_kv,
                     (long)ld_o, scale, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}

extern "C" int tfpp_small_attn_bwd(const float* q, const float* k, const float* v, const float* p_save, const float* d_o, float* dq, float* dk,
                                   float* dv, int B, int nh, int tq, int tk, int d, int64_t ld_q, int64_t ld_kv, int64_t ld_o, float scale,
                                   float p_drop, uint64_t seed, const uint64_t* seed_offset, void* stream) {
  if (!q || !k || !v || !p_save || !d_o || !dq || !dk || !dv || B < 1 || nh < 1 || !tfpp_small_attn_supported(tq, tk, d) || p_drop < 0.f ||
      p_drop >= 1.f)
    return TFPP_EINVAL;
  const float inv_keep = p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f;
  hipLaunchKernelGGL(small_attn_bwd_kernel, dim3(B * nh), dim3(256), 0, (hipStream_t)stream, q, k, v, p_save, d_o, dq, dk, dv, nh, tq, tk, d, (long)ld_q,
                     (long)ld_kv, (long)ld_o, scale, p_drop, inv_keep, (unsigned long long)seed, (const unsigned long long*)seed_offset);
  TFPP_CHECK_LAUNCH();
  return 0;
}
