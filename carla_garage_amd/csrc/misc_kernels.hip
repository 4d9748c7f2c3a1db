// GRU waypoint decoder, fused loss+gradient kernels and the AdamW(amsgrad) optimizer step.
#include "common.h"
#include "../../include/tfpp.h"

// ---------------------------------------------------------------------------------------------------------------
// GRU (team_code/model.py:857-867): one block per sample, hidden state in LDS, T sequential steps.
// torch.nn.GRU gates:  r = s(gi_r + gh_r)  z = s(gi_z + gh_z)  n = tanh(gi_n + r*gh_n)  h' = (1-z)*n + z*h
// save[b][t] = {r, z, n, h'} (4*H floats); out[b][t][:] = cumsum_t(W_dec h'_t + b_dec)
// ---------------------------------------------------------------------------------------------------------------
#define GRU_MAXH 64
__global__ void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0, const float* __restrict__ w_hh,
                               const float* __restrict__ b_hh, const float* __restrict__ w_dec, const float* __restrict__ b_dec,
                               float* __restrict__ save, float* __restrict__ out, int T, int H) {
  __shared__ float h[GRU_MAXH], gh[3 * GRU_MAXH], cum[2];
  const int b = blockIdx.x, tid = threadIdx.x;  // 3*H threads
  if (tid < H) h[tid] = h0[(size_t)b * H + tid];
  if (tid < 2) cum[tid] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    float s = b_hh[tid];
    for (int k = 0; k < H; ++k) s += w_hh[(size_t)tid * H + k] * h[k];
    gh[tid] = s;
    __syncthreads();
    if (tid < H) {
      const float* g = gi + ((size_t)b * T + t) * 3 * H;
      const float r = 1.f / (1.f + __expf(-(g[tid] + gh[tid])));
      const float z = 1.f / (1.f + __expf(-(g[H + tid] + gh[H + tid])));
      const float n = tanhf(g[2 * H + tid] + r * gh[2 * H + tid]);
      const float hn = (1.f - z) * n + z * h[tid];
      float* sv = save + ((size_t)b * T + t) * 4 * H;
      sv[tid] = r; sv[H + tid] = z; sv[2 * H + tid] = n; sv[3 * H + tid] = hn;
      h[tid] = hn;
    }
    __syncthreads();
    if (tid < 2) {
      float s2 = b_dec[tid];
      for (int k = 0; k < H; ++k) s2 += w_dec[tid * H + k] * h[k];
      cum[tid] += s2;
      out[((size_t)b * T + t) * 2 + tid] = cum[tid];
    }
    __syncthreads();
  }
}

extern "C" int tfpp_gru_fwd(const float* gi, const float* h0, const float* w_hh, const float* b_hh, const float* w_dec, const float* b_dec,
                            float* save, float* out, int B, int T, int H, void* stream) {
  if (!gi || !h0 || !w_hh || !b_hh || !w_dec || !b_dec || !save || !out || H > GRU_MAXH) return TFPP_EINVAL;
  hipLaunchKernelGGL(gru_fwd_kernel, dim3(B), dim3(3 * H), 0, (hipStream_t)stream, gi, h0, w_hh, b_hh, w_dec, b_dec, save, out, T, H);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Backward.  Round 3: the parameter gradients no longer go through atomics (120 adds per address per step, on the first node of the planning head's
// backward chain: 0.3 ms): every thread keeps "its" row of dW_hh in 64 registers over the T steps, W_hh lies in LDS (row stride H + 1: conflict-free
// for the row-wise and the column-wise product), each sample writes ONE partial image  part[b] = {dW_hh[3H][H], db_hh[3H], dW_dec[2][H], db_dec[2]}
// and gru_bwd_reduce_kernel adds the B images in a fixed order (bit-reproducible; it runs on the weight-gradient lane).
#define GRU_PART(H) (3 * (H) * (H) + 3 * (H) + 2 * (H) + 2)
__global__ void __launch_bounds__(3 * GRU_MAXH)
gru_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ save, const float* __restrict__ h0,
               const float* __restrict__ w_hh, const float* __restrict__ b_hh, const float* __restrict__ w_dec,
               float* __restrict__ dgi, float* __restrict__ dh0, float* __restrict__ part, int T, int H) {
  __shared__ float w[3 * GRU_MAXH * (GRU_MAXH + 1)];
  __shared__ float dh[GRU_MAXH], hp[GRU_MAXH], dgh[3 * GRU_MAXH], ghn[GRU_MAXH], dcum[2], dhn[GRU_MAXH];
  const int b = blockIdx.x, tid = threadIdx.x, ldw = H + 1;  // 3*H threads
  for (int i = tid; i < 3 * H * H; i += 3 * H) w[(i / H) * ldw + (i % H)] = w_hh[i];
  float acc[GRU_MAXH];
#pragma unroll
  for (int k = 0; k < GRU_MAXH; ++k) acc[k] = 0.f;
  float acc_b = 0.f, acc_d0 = 0.f, acc_d1 = 0.f, acc_bd = 0.f;
  if (tid < H) dh[tid] = 0.f;
  if (tid < 2) dcum[tid] = 0.f;
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const float* sv = save + ((size_t)b * T + t) * 4 * H;
    // reverse cumsum: gradient of the per-step decoder output o_t is sum_{tau>=t} dout[tau]
    if (tid < 2) dcum[tid] += dout[((size_t)b * T + t) * 2 + tid];
    if (tid < H) hp[tid] = (t > 0) ? save[((size_t)b * T + t - 1) * 4 * H + 3 * H + tid] : h0[(size_t)b * H + tid];
    __syncthreads();
    if (tid < H) {
      const float hn = sv[3 * H + tid];
      dhn[tid] = dh[tid] + w_dec[tid] * dcum[0] + w_dec[H + tid] * dcum[1];
      acc_d0 += dcum[0] * hn;
      acc_d1 += dcum[1] * hn;
      // recompute gh_n = W_hn h_{t-1} + b_hn
      float s = b_hh[2 * H + tid];
      const float* wr = w + (2 * H + tid) * ldw;
      for (int k = 0; k < H; ++k) s += wr[k] * hp[k];
      ghn[tid] = s;
    }
    if (tid < 2) acc_bd += dcum[tid];
    __syncthreads();
    if (tid < H) {
      const float r = sv[tid], z = sv[H + tid], n = sv[2 * H + tid];
      const float d = dhn[tid];
      const float dn = d * (1.f - z);
      const float dz = d * (hp[tid] - n);
      const float dpn = dn * (1.f - n * n);
      const float dr = dpn * ghn[tid];
      const float dpz = dz * z * (1.f - z);
      const float dpr = dr * r * (1.f - r);
      float* g = dgi + ((size_t)b * T + t) * 3 * H;
      g[tid] = dpr; g[H + tid] = dpz; g[2 * H + tid] = dpn;
      dgh[tid] = dpr; dgh[H + tid] = dpz; dgh[2 * H + tid] = dpn * r;
      dh[tid] = d * z;  // direct path h_{t-1} -> h_t
    }
    __syncthreads();
    // parameter gradients of the recurrent projection (registers) and the gradient w.r.t. h_{t-1}
    const float gme = dgh[tid];
    acc_b += gme;
#pragma unroll
    for (int k = 0; k < GRU_MAXH; ++k)
      if (k < H) acc[k] += gme * hp[k];
    if (tid < H) {
      float s = 0.f;
      for (int j = 0; j < 3 * H; ++j) s += dgh[j] * w[j * ldw + tid];
      dh[tid] += s;
    }
    __syncthreads();
  }
  if (tid < H) dh0[(size_t)b * H + tid] = dh[tid];
  // the partial image of this sample: rows of dW_hh go through LDS (the W_hh copy is dead) so that the global stores are coalesced
#pragma unroll
  for (int k = 0; k < GRU_MAXH; ++k)
    if (k < H) w[tid * ldw + k] = acc[k];
  __syncthreads();
  float* pb = part + (size_t)b * GRU_PART(H);
  for (int i = tid; i < 3 * H * H; i += 3 * H) pb[i] = w[(i / H) * ldw + (i % H)];
  pb[3 * H * H + tid] = acc_b;
  if (tid < H) {
    pb[3 * H * H + 3 * H + tid] = acc_d0;
    pb[3 * H * H + 3 * H + H + tid] = acc_d1;
  }
  if (tid < 2) pb[3 * H * H + 5 * H + tid] = acc_bd;
}

// gradient destinations += sum over the B partial images, samples in index order
__global__ void gru_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw_hh, float* __restrict__ db_hh,
                                      float* __restrict__ dw_dec, float* __restrict__ db_dec, int B, int H) {
  const int P = GRU_PART(H), i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[(size_t)b * P + i];
  const int o1 = 3 * H * H, o2 = o1 + 3 * H, o3 = o2 + 2 * H;
  if (i < o1) dw_hh[i] += s;
  else if (i < o2) db_hh[i - o1] += s;
  else if (i < o3) dw_dec[i - o2] += s;
  else db_dec[i - o3] += s;
}

extern "C" int tfpp_gru_bwd_partial_floats(int B, int H) { return B * GRU_PART(H); }

extern "C" int tfpp_gru_bwd_reduce(const float* partial, float* dw_hh, float* db_hh, float* dw_dec, float* db_dec, int B, int H, void* stream) {
  if (!partial || !dw_hh || !db_hh || !dw_dec || !db_dec || H > GRU_MAXH || B < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(gru_bwd_reduce_kernel, dim3((GRU_PART(H) + 255) / 256), dim3(256), 0, (hipStream_t)stream, partial, dw_hh, db_hh, dw_dec, db_dec, B, H);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// dw_* / db_* may all be null: the caller then runs tfpp_gru_bwd_reduce itself (on another stream, once `partial` is complete)
extern "C" int tfpp_gru_bwd(const float* dout, const float* save, const float* h0, const float* w_hh, const float* b_hh, const float* w_dec,
                            float* dgi, float* dh0, float* partial, float* dw_hh, float* db_hh, float* dw_dec, float* db_dec, int B, int T, int H,
                            void* stream) {
  if (!dout || !save || !h0 || !w_hh || !b_hh || !w_dec || !dgi || !dh0 || !partial || H > GRU_MAXH || B < 1) return TFPP_EINVAL;
  const bool any = dw_hh || db_hh || dw_dec || db_dec;
  if (any && !(dw_hh && db_hh && dw_dec && db_dec)) return TFPP_EINVAL;
  hipLaunchKernelGGL(gru_bwd_kernel, dim3(B), dim3(3 * H), 0, (hipStream_t)stream, dout, save, h0, w_hh, b_hh, w_dec, dgi, dh0, partial, T, H);
  TFPP_CHECK_LAUNCH();
  if (any) return tfpp_gru_bwd_reduce(partial, dw_hh, db_hh, dw_dec, db_dec, B, H, stream);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm1d(1, affine=False) on the ego speed (model.py:216,311): tiny, one block
// ---------------------------------------------------------------------------------------------------------------
__global__ void bn1d_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ rm, float* __restrict__ rv,
                                   long long* __restrict__ nbt, int B, int training, float momentum, float eps) {
  __shared__ float stat[2];
  if (threadIdx.x == 0) {
    float m, v;
    if (training) {
      double s = 0.0, q = 0.0;
      for (int i = 0; i < B; ++i) s += x[i];
      m = (float)(s / B);
      for (int i = 0; i < B; ++i) { const double d = x[i] - m; q += d * d; }
      v = (float)(q / B);
      rm[0] = (1.f - momentum) * rm[0] + momentum * m;
      rv[0] = (1.f - momentum) * rv[0] + momentum * (float)(B > 1 ? q / (B - 1) : q);
      if (nbt) *nbt += 1;
    } else {
      m = rm[0]; v = rv[0];
    }
    stat[0] = m; stat[1] = 1.f / sqrtf(v + eps);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += blockDim.x) y[i] = (x[i] - stat[0]) * stat[1];
}

extern "C" int tfpp_bn1d_scalar(const float* x, float* y, float* running_mean, float* running_var, int64_t* nbt, int B, int training,
                                float momentum, float eps, void* stream) {
  if (!x || !y || !running_mean || !running_var) return TFPP_EINVAL;
  hipLaunchKernelGGL(bn1d_scalar_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, y, running_mean, running_var, (long long*)nbt, B, training,
                     momentum, eps);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// losses.  pred is [rows, ld] NHWC (C real classes / channels); labels keep the reference's layouts.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = (threadIdx.x < (blockDim.x >> 6)) ? sm[threadIdx.x] : 0.f;
  if (w == 0) r = wave_sum(r);
  __syncthreads();
  return r;  // valid in wave 0
}

// effective label: -1 (ignored) where the visibility mask is 0 (model.py:427-429), else the label
__device__ __forceinline__ long long ce_label(const long long* __restrict__ label, const float* __restrict__ vis, long i, long HW) {
  long long l = label[i];
  if (vis && vis[i % HW] == 0.f) l = -1;
  return l;
}

// The last workgroup of a loss kernel: sum of the gridDim.x published partials in index order (thread t adds t, t + 256, ...; then the fixed
// tree of block_sum_256).  Valid in thread 0.
__device__ __forceinline__ float grid_sum_partials(const float* __restrict__ partials, float* sm) {
  float s = 0.f;
  for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) s += grid_fetch(partials + i);
  return block_sum_256(s, sm);
}

template <typename T, bool VROW> __device__ __forceinline__ void ce_store_row(T* d, const float* g, int ld) {
  if (VROW) {
    constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
    for (int k = 0; k < 16 / VEC; ++k)
      if (k * VEC < ld) store_vec<T>(d + k * VEC, g + k * VEC);
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (c < ld) d[c] = ElemTraits<T>::from_f(g[c]);
  }
}

// pass 1 of cross entropy: ws[0] = sum_i class_weight[label_i] over non-ignored rows (fixed-order grid sum: common.h)
__global__ void ce_norm_kernel(const long long* __restrict__ label, const float* __restrict__ cw, const float* __restrict__ vis, long HW,
                               float* __restrict__ ws, long rows, float* __restrict__ scratch) {
  __shared__ float sm[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
    const long long l = ce_label(label, vis, i, HW);
    if (l >= 0) s += cw ? cw[l] : 1.f;
  }
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) grid_publish(scratch + TFPP_GRIDSUM_TICKETS + blockIdx.x, s);
  if (!grid_last_ticket(reinterpret_cast<unsigned*>(scratch), gridDim.x)) return;
  s = grid_sum_partials(scratch + TFPP_GRIDSUM_TICKETS, sm);
  if (threadIdx.x == 0) ws[0] = s;
}

// pass 2: loss and gradient.  denominator: pix_weight mode -> (*denom + eps)   else ws[0] (weighted mean)
// VROW: a row (ld elements) is a whole number of aligned 16-byte vectors -> vector loads of the logits and vector stores of the gradient
// (the full-resolution semantic map is 3.1 M rows of 8 bf16: 16 scalar loads per row made this kernel ALU / issue bound)
template <typename T, bool VROW>
__global__ void ce_loss_kernel(const T* __restrict__ pred, const long long* __restrict__ label, const float* __restrict__ cw,
                               const float* __restrict__ vis, const float* __restrict__ pix_weight, long pw_bstride, long HW,
                               const float* __restrict__ denom, float denom_eps, const float* __restrict__ ws, float weight,
                               float* __restrict__ loss_out, T* __restrict__ dpred, long rows, int C, int ld, float smoothing,
                               float focal_gamma, float* __restrict__ scratch) {
  __shared__ float sm[4];
  // label smoothing (nn.CrossEntropyLoss(weight, label_smoothing), model.py:252-265): per row (1 - a) w[y] nll(y) + a / C sum_c w[c] nll(c),
  // normalised like the unsmoothed loss by sum_i w[y_i]
  float wsum = 0.f;
  if (smoothing > 0.f)
    for (int c = 0; c < C; ++c) wsum += cw ? cw[c] : 1.f;
  // focal form (focal_gamma >= 0; team_code/focal_loss.py:75-103): mean over ALL rows of alpha[y] (1 - p_y)^gamma (-log p_y)
  const bool focal = focal_gamma >= 0.f;
  const float den = focal ? (float)rows : (pix_weight ? (denom[0] + denom_eps) : ws[0]);
  const float inv = den > 0.f ? 1.f / den : 0.f;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
    const T* p = pred + (size_t)i * ld;
    const long long l = ce_label(label, vis, i, HW);
    float v[16];
    float mx = -3.0e38f;
    if (VROW) {
      constexpr int VEC = ElemTraits<T>::VEC;
#pragma unroll
      for (int k = 0; k < 16 / VEC; ++k)
        if (k * VEC < ld) load_vec<T>(p + k * VEC, v + k * VEC);
#pragma unroll
      for (int c = 0; c < 16; ++c) { v[c] = (c < C) ? v[c] : -3.0e38f; mx = fmaxf(mx, v[c]); }
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) { v[c] = (c < C) ? ElemTraits<T>::to_f(p[c < ld ? c : 0]) : -3.0e38f; mx = fmaxf(mx, v[c]); }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) { v[c] = (c < C) ? __expf(v[c] - mx) : 0.f; s += v[c]; }
    float w = 0.f;
    if (l >= 0) {
      w = cw ? cw[l] : 1.f;
      if (pix_weight) w *= pix_weight[(i / HW) * pw_bstride + (i % HW)];
    }
    const float invs = 1.f / s;
    float smooth_nll = 0.f;
    if (l >= 0) {
      float pl = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) pl = (c == (int)l) ? v[c] * invs : pl;
      if (focal) {
        // f(p) = -w (1 - p)^g log p;  df/dz_c = w [g (1 - p)^(g-1) p log p - (1 - p)^g] (delta_cy - softmax_c)
        const float lp = __logf(fmaxf(pl, 1e-38f)), om = fmaxf(1.f - pl, 0.f);
        const float pg = focal_gamma == 0.f ? 1.f : __powf(om, focal_gamma);
        acc += -w * pg * lp;
        if (dpred) {
          const float pg1 = focal_gamma == 0.f ? 0.f : focal_gamma * (focal_gamma == 1.f ? 1.f : __powf(om, focal_gamma - 1.f));
          const float coef = weight * inv * w * (pg1 * pl * lp - pg);
          float gout[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) gout[c] = c < C ? coef * (((c == (int)l) ? 1.f : 0.f) - v[c] * invs) : 0.f;
          ce_store_row<T, VROW>(dpred + (size_t)i * ld, gout, ld);
        }
        continue;
      }
      acc += -(1.f - smoothing) * w * __logf(fmaxf(pl, 1e-38f));
      if (smoothing > 0.f) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c < C) smooth_nll += -(cw ? cw[c] : 1.f) * __logf(fmaxf(v[c] * invs, 1e-38f));
        acc += smoothing / (float)C * smooth_nll;
      }
    }
    if (dpred) {
      const float gs = weight * inv;
      const float wp = (l >= 0) ? (1.f - smoothing) * w + smoothing / (float)C * wsum : 0.f;  // coefficient of softmax(z)_c
      float gout[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        float g = 0.f;
        if (c < C && l >= 0)
          g = gs * (wp * v[c] * invs - ((c == (int)l) ? (1.f - smoothing) * w : 0.f) - smoothing / (float)C * (cw ? cw[c] : 1.f));
        gout[c] = g;
      }
      ce_store_row<T, VROW>(dpred + (size_t)i * ld, gout, ld);
    }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) grid_publish(scratch + TFPP_GRIDSUM_TICKETS + blockIdx.x, acc);
  if (!grid_last_ticket(reinterpret_cast<unsigned*>(scratch), gridDim.x)) return;
  acc = grid_sum_partials(scratch + TFPP_GRIDSUM_TICKETS, sm);
  if (threadIdx.x == 0) loss_out[0] += acc * inv;  // (one writer: the losses accumulate into their slot)
}

extern "C" int tfpp_ce_loss(const void* pred, const int64_t* label, const float* class_weight, const float* vis_mask, const float* pix_weight,
                            int64_t pw_bstride, int64_t HW, const float* denom, float denom_eps, float weight, float* loss_out, void* dpred,
                            float* ws, float* scratch, int64_t rows, int C, int ld, float smoothing, float focal_gamma, int dtype, void* stream) {
  if (focal_gamma >= 0.f && (pix_weight || smoothing > 0.f || vis_mask)) return TFPP_EINVAL;
  if (!pred || !label || !loss_out || !ws || !scratch || C > 16 || ld > 16 || ld < C || HW < 1 || (pix_weight && !denom)) return TFPP_EINVAL;
  if (smoothing < 0.f || smoothing >= 1.f || (smoothing > 0.f && pix_weight)) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long blocks = (rows + 255) / 256;
  if (blocks > 1024) blocks = 1024;  // (one partial + one ticket per workgroup: 1024 x 256 threads stream at the HBM rate, 4096 tickets on one address cost more than they buy)
  if (blocks < 1) blocks = 1;
  const int vec = dtype == TFPP_F32 ? 4 : 8;
  const bool vrow = ld % vec == 0 && !((uintptr_t)pred & 15) && !((uintptr_t)dpred & 15);
  if (!pix_weight && focal_gamma < 0.f)
    hipLaunchKernelGGL(ce_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const long long*)label, class_weight, vis_mask, (long)HW, ws, (long)rows, scratch);
#define CE_LAUNCH(TT, VR) hipLaunchKernelGGL((ce_loss_kernel<TT, VR>), dim3((unsigned)blocks), dim3(256), 0, st, (const TT*)pred, (const long long*)label, class_weight, vis_mask, pix_weight, (long)pw_bstride, (long)HW, denom, denom_eps, ws, weight, loss_out, (TT*)dpred, (long)rows, C, ld, smoothing, focal_gamma, scratch)
  if (dtype == TFPP_F32) { if (vrow) CE_LAUNCH(float, true); else CE_LAUNCH(float, false); }
  else { if (vrow) CE_LAUNCH(bf16_t, true); else CE_LAUNCH(bf16_t, false); }
#undef CE_LAUNCH
  TFPP_CHECK_LAUNCH();
  return 0;
}

// regression-type losses over logical elements (b, c, pix): pred at [(b*HW+pix)*ld + c]; target (NCHW) at [(b*C+c)*HW+pix];
// elem weight (NCHW with wC channels) at [(b*wC + (w_bcast?0:c))*HW + pix].
// kind 0: L1   1: smooth-L1 (beta 1)   2: gaussian focal loss on sigmoid outputs (transfuser_utils.py:341-364)
// loss = sum(...)/den, den = denom ? (*denom + denom_eps) * denom_mul : B*C*HW
__device__ __forceinline__ float reg_loss_elem(int kind, float p, float t, float w, float& g) {
  if (kind == 0) {
    const float d = p - t;
    g = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * w;
    return fabsf(d) * w;
  }
  if (kind == 1) {
    const float d = p - t, ad = fabsf(d);
    g = (ad < 1.f ? d : (d > 0.f ? 1.f : -1.f)) * w;
    return (ad < 1.f ? 0.5f * d * d : ad - 0.5f) * w;
  }
  const float eps = 1e-12f;
  if (t == 1.f) {
    const float om = 1.f - p, lg = __logf(p + eps);
    g = -om * om / (p + eps) + 2.f * om * lg;
    return -lg * om * om;
  }
  const float omt = 1.f - t, nw = omt * omt * omt * omt, lg = __logf(1.f - p + eps);
  g = (p * p / (1.f - p + eps) - 2.f * p * lg) * nw;
  return -lg * p * p * nw;
}

// One thread per pixel ROW (b, pix): the ld stored channels of a row are one or two aligned 16-byte vectors (VROW) or scalar loads; one
// division per row instead of three 64-bit divisions per element (the full-resolution depth map is 3.1 M rows x 8 bf16).
template <typename T, bool VROW>
__global__ void reg_loss_kernel(const T* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ ew, int wC, int w_bcast,
                                const float* __restrict__ denom, float denom_eps, float denom_mul, float weight, float* __restrict__ loss_out,
                                T* __restrict__ dpred, int B, int C, long HW, long ld, int kind, float* __restrict__ scratch) {
  __shared__ float sm[4];
  constexpr int VEC = ElemTraits<T>::VEC;
  const long nrows = (long)B * HW;
  const float den = denom ? (denom[0] + denom_eps) * denom_mul : (float)((double)B * C * HW);
  const float inv = 1.f / den;
  const float gscale = weight * inv;
  float acc = 0.f;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long)gridDim.x * blockDim.x) {
    const long b = r / HW, pix = r - b * HW;
    const T* prow = pred + (size_t)r * ld;
    T* drow = dpred ? dpred + (size_t)r * ld : nullptr;
    for (int c0 = 0; c0 < (int)ld; c0 += VEC) {
      float p[VEC], g[VEC];
      if (VROW) {
        load_vec<T>(prow + c0, p);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) p[e] = c0 + e < (int)ld ? ElemTraits<T>::to_f(prow[c0 + e]) : 0.f;
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const int c = c0 + e;
        g[e] = 0.f;
        if (c < C) {
          const float t = target[((size_t)b * C + c) * HW + pix];
          const float w = ew ? ew[((size_t)b * wC + (w_bcast ? 0 : c)) * HW + pix] : 1.f;
          acc += reg_loss_elem(kind, p[e], t, w, g[e]);
          g[e] *= gscale;
        }
      }
      if (drow) {
        if (VROW) {
          store_vec<T>(drow + c0, g);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (c0 + e < (int)ld) drow[c0 + e] = ElemTraits<T>::from_f(g[e]);
        }
      }
    }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) grid_publish(scratch + TFPP_GRIDSUM_TICKETS + blockIdx.x, acc);
  if (!grid_last_ticket(reinterpret_cast<unsigned*>(scratch), gridDim.x)) return;
  acc = grid_sum_partials(scratch + TFPP_GRIDSUM_TICKETS, sm);
  if (threadIdx.x == 0) loss_out[0] += acc * inv;  // (one writer: the losses accumulate into their slot)
}

extern "C" int tfpp_reg_loss(const void* pred, const float* target, const float* elem_weight, int wC, int w_bcast, const float* denom,
                             float denom_eps, float denom_mul, float weight, float* loss_out, void* dpred, float* scratch, int B, int C, int64_t HW,
                             int64_t ld, int kind, int dtype, void* stream) {
  if (!pred || !target || !loss_out || !scratch || ld < C) return TFPP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long nrows = (long)B * HW;
  long blocks = (nrows + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  const int vec = dtype == TFPP_F32 ? 4 : 8;
  const bool vrow = ld % vec == 0 && !((uintptr_t)pred & 15) && !((uintptr_t)dpred & 15);
#define REG_LAUNCH(TT, VR) hipLaunchKernelGGL((reg_loss_kernel<TT, VR>), dim3((unsigned)blocks), dim3(256), 0, st, (const TT*)pred, target, elem_weight, wC, w_bcast, denom, denom_eps, denom_mul, weight, loss_out, (TT*)dpred, B, C, (long)HW, (long)ld, kind, scratch)
  if (dtype == TFPP_F32) { if (vrow) REG_LAUNCH(float, true); else REG_LAUNCH(float, false); }
  else { if (vrow) REG_LAUNCH(bf16_t, true); else REG_LAUNCH(bf16_t, false); }
#undef REG_LAUNCH
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Two waypoint hypotheses, the better one per sample is trained (config.multi_wp_output, model.py:401-408): l_h[b] = mean_e |pair[b,h,e] - label[b,e]|,
// loss = mean_b min(l_0, l_1); the arg-min (0 / 1, the first on a tie as torch.min) is the label of the path-selection logit.  One wave per
// sample; fp32 (planning head).  dpair: d(weight * loss) / d pair, zero for the hypothesis that lost.
#define TFPP_MULTI_WP_MAX_B 1024
__global__ void min_l1_pair_loss_kernel(const float* __restrict__ pair, const float* __restrict__ label, float weight, float* __restrict__ loss_out,
                                        float* __restrict__ dpair, float* __restrict__ sel_label, int B, int n) {
  __shared__ float best[TFPP_MULTI_WP_MAX_B];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const float gs = weight / ((float)n * (float)B);
  for (int b = w; b < B; b += nw) {
    const float* p0 = pair + (size_t)b * 2 * n;
    const float* p1 = p0 + n;
    const float* l = label + (size_t)b * n;
    float a0 = 0.f, a1 = 0.f;
    for (int e = lane; e < n; e += 64) {
      a0 += fabsf(p0[e] - l[e]);
      a1 += fabsf(p1[e] - l[e]);
    }
    a0 = wave_sum(a0) / (float)n;
    a1 = wave_sum(a1) / (float)n;
    const int pick = a1 < a0 ? 1 : 0;
    if (lane == 0) {
      best[b] = pick ? a1 : a0;
      sel_label[b] = (float)pick;
    }
    if (dpair) {
      float* d0 = dpair + (size_t)b * 2 * n;
      for (int e = lane; e < 2 * n; e += 64) {
        const int h = e >= n, ee = e - h * n;
        const float df = p0[e] - l[ee];  // (p0 + n + ee == p1 + ee)
        d0[e] = h == pick ? (df > 0.f ? gs : (df < 0.f ? -gs : 0.f)) : 0.f;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += best[b];
    loss_out[0] += s / (float)B;
  }
}
extern "C" int tfpp_min_l1_pair_loss(const float* pair, const float* label, float weight, float* loss_out, float* dpair, float* sel_label, int B,
                                     int n, void* stream) {
  if (!pair || !label || !loss_out || !sel_label || B < 1 || B > TFPP_MULTI_WP_MAX_B || n < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(min_l1_pair_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pair, label, weight, loss_out, dpair, sel_label, B, n);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// nn.BCEWithLogitsLoss() on one logit per sample (model.py:266-267,409-411): mean_b [max(x, 0) - x y + log(1 + exp(-|x|))]; x = logit[b * ld],
// the other ld - 1 stored channels are padding and get a zero gradient.
__global__ void bce_logits_loss_kernel(const float* __restrict__ logit, int ld, const float* __restrict__ y, float weight, float* __restrict__ loss_out,
                                       float* __restrict__ dlogit, int B) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float x = logit[(size_t)b * ld], t = y[b];
    acc += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    if (dlogit) {
      const float sg = x >= 0.f ? 1.f / (1.f + expf(-x)) : expf(x) / (1.f + expf(x));
      dlogit[(size_t)b * ld] = weight * (sg - t) / (float)B;
      for (int c = 1; c < ld; ++c) dlogit[(size_t)b * ld + c] = 0.f;
    }
  }
  acc = block_sum_256(acc, sm);
  if (threadIdx.x == 0) loss_out[0] += acc / (float)B;
}
extern "C" int tfpp_bce_logits_loss(const float* logit, int ld, const float* y, float weight, float* loss_out, float* dlogit, int B, void* stream) {
  if (!logit || !y || !loss_out || B < 1 || ld < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(bce_logits_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logit, ld, y, weight, loss_out, dlogit, B);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// out[0] = sum(x[0..n))  (avg_factor.sum(), center_net.py:98)
__global__ void sum_f32_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
  __shared__ float sm[4];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  s = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = s;
}
extern "C" int tfpp_sum_f32(const float* x, float* out, int64_t n, void* stream) {
  if (!x || !out) return TFPP_EINVAL;
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// AdamW with amsgrad (torch.optim.AdamW semantics, train.py:529-531) over a flat fp32 arena
// ---------------------------------------------------------------------------------------------------------------
__global__ void adamw_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                     float* __restrict__ vmax, long n, float lr, float beta1, float beta2, float eps, float wd_all, float bc1,
                                     float bc2_sqrt, float grad_scale, const unsigned* __restrict__ no_decay) {
  // no_decay (nullable): one bit per group of 4 consecutive elements of the arena (every parameter starts on a multiple of 4): set = this
  // parameter is in the weight_decay = 0 group of create_optimizer_groups (model.py:556-632: biases, norms, embeddings, queries)
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    const long q = i >> 2;
    const float wd = (no_decay && ((no_decay[q >> 5] >> (q & 31)) & 1u)) ? 0.f : wd_all;
    if (i + 3 < n) {
      float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i), mm = *reinterpret_cast<float4*>(m + i),
             vv = *reinterpret_cast<float4*>(v + i), xx = *reinterpret_cast<float4*>(vmax + i);
      float* pa = &pp.x; float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x; float* xa = &xx.x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gr = ga[e] * grad_scale;
        pa[e] *= (1.f - lr * wd);
        ma[e] = beta1 * ma[e] + (1.f - beta1) * gr;
        va[e] = beta2 * va[e] + (1.f - beta2) * gr * gr;
        xa[e] = fmaxf(xa[e], va[e]);
        pa[e] -= (lr / bc1) * ma[e] / (sqrtf(xa[e]) / bc2_sqrt + eps);
      }
      *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
      *reinterpret_cast<float4*>(vmax + i) = xx;
    } else {
      for (long j = i; j < n; ++j) {
        const float gr = g[j] * grad_scale;
        float pj = p[j] * (1.f - lr * wd);
        const float mj = beta1 * m[j] + (1.f - beta1) * gr;
        const float vj = beta2 * v[j] + (1.f - beta2) * gr * gr;
        const float xj = fmaxf(vmax[j], vj);
        pj -= (lr / bc1) * mj / (sqrtf(xj) / bc2_sqrt + eps);
        p[j] = pj; m[j] = mj; v[j] = vj; vmax[j] = xj;
      }
    }
  }
}

static int launch_adamw(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, const unsigned* no_decay, void* stream);

extern "C" int tfpp_adamw_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, float grad_scale, void* stream) {
  return launch_adamw(p, g, m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}

// Two parameter groups (team_code/train.py:522-523, use_optim_groups): elements whose bit in no_decay_bits (one bit per 4 elements, p must be the
// start of the arena the bits were built for) is set take weight_decay = 0, the others `weight_decay`.
extern "C" int tfpp_adamw_amsgrad_groups(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, int step, float grad_scale, const uint32_t* no_decay_bits, void* stream) {
  if (!no_decay_bits) return TFPP_EINVAL;
  return launch_adamw(p, g, m, v, vmax, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, no_decay_bits, stream);
}

static int launch_adamw(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float grad_scale, const unsigned* no_decay, void* stream) {
  if (!p || !g || !m || !v || !vmax || step < 1) return TFPP_EINVAL;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) return TFPP_EINVAL;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, (long)n, lr, beta1, beta2,
                     eps, weight_decay, bc1, bc2_sqrt, grad_scale, no_decay);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// scratch of the fixed-order grid sums (common.h): 64 ticket counters + the partials of the largest user (LayerNorm parameter gradients:
// 64 row blocks x 2 x C <= 3072; the loss kernels publish <= 1024 partials).  Zero before the first use; one buffer per stream.
extern "C" int tfpp_gridsum_scratch_floats(void) { return TFPP_GRIDSUM_TICKETS + 64 * 2 * 3072; }

extern "C" int tfpp_version(void) { return TFPP_ABI_VERSION; }

// Hash of the kernel sources this binary was built from (passed in by carla_garage_amd/_lib.py::build): the loader compares it
// with the sources next to it, so a stale libtfpp_hip.so (git-ignored, shipped out of band) is rejected instead of tested.
#ifndef TFPP_SOURCE_HASH
#define TFPP_SOURCE_HASH 0ULL
#endif
extern "C" int tfpp_source_hash(uint64_t* out) {
  if (!out) return TFPP_EINVAL;
  *out = (uint64_t)TFPP_SOURCE_HASH;
  return 0;
}

extern "C" int tfpp_struct_sizes(int* out, int n) {
  if (!out || n < 6) return TFPP_EINVAL;
  out[0] = (int)sizeof(tfpp_conv_params);
  out[1] = (int)sizeof(tfpp_wgrad_params);
  out[2] = (int)sizeof(tfpp_bgemm_params);
  out[3] = (int)sizeof(tfpp_pack_desc);
  out[4] = (int)sizeof(tfpp_attn_params);
  out[5] = (int)sizeof(tfpp_bn_rows);
  return 6;
}

// order-independent 64-bit hash of a buffer (debugging aid: which tensor differs between two replays of one hipGraph?)
__global__ void hash_words_kernel(const unsigned int* __restrict__ w, long n, unsigned long long* __restrict__ slot) {
  unsigned long long h = 0ull;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned long long x = ((unsigned long long)w[i] << 32 | (unsigned long long)(unsigned int)i) ^ ((unsigned long long)(i >> 32) * 0x9E3779B97F4A7C15ull);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    h += x;
  }
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
  if ((threadIdx.x & 63) == 0 && h) atomicAdd(slot, h);
}
extern "C" int tfpp_hash_words(const void* p, int64_t bytes, uint64_t* slot, void* stream) {
  if (!p || !slot || bytes < 0 || (bytes & 3)) return TFPP_EINVAL;
  const long n = bytes >> 2;
  if (n == 0) return 0;
  long blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(hash_words_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned int*)p, n, (unsigned long long*)slot);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// debugging aid (tools/lane_timeline.py): *slot = the device's constant-rate wall clock (100 MHz) when this launch runs -- a time stamp INSIDE a
// captured step, on whatever lane it was issued
__global__ void stamp_kernel(unsigned long long* slot) { *slot = wall_clock64(); }
extern "C" int tfpp_stamp(uint64_t* slot, void* stream) {
  if (!slot) return TFPP_EINVAL;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// *p += 1 : per-step counter added to every dropout seed (captured in the training-step hipGraph)
__global__ void inc_u64_kernel(unsigned long long* p) { *p += 1ull; }
extern "C" int tfpp_inc_u64(uint64_t* p, void* stream) {
  if (!p) return TFPP_EINVAL;
  hipLaunchKernelGGL(inc_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)p);
  TFPP_CHECK_LAUNCH();
  return 0;
}

// Completion signal between a kernel node INSIDE a captured hipGraph and a stream OUTSIDE of it (carla_garage_amd/buckets.py).  ROCm 7.x offers
// no host-side primitive for that (tools/graph_external_event_test.py: PyTorch refuses external events on ROCm, hipEventRecordWithFlags(...,
// hipEventRecordExternal) returns hipErrorInvalidValue inside a capture, hipMallocSignalMemory is not available), so the signal is a counter
// in device memory: tfpp_signal_set (below) is a one-thread kernel node behind the kernels that complete a gradient bucket -- it raises the
// word to the serial number of the running pass --, tfpp_signal_wait is a one-wave kernel on the collective's stream that polls it with a device-scope atomic (coherent across the XCDs'
// L2s, ~1 poll per microsecond, s_sleep in between) until it reaches `value`; the kernel boundary behind it is the acquire for whatever the
// next kernel on that stream (the RCCL all-reduce) reads.  A wait that sees nothing for `timeout_ms` gives up and counts in *timeouts
// (a stuck stream must never hang the GPU box); the caller checks that word at its next synchronisation point.
__global__ void signal_wait_kernel(unsigned long long* sig, unsigned long long value, unsigned long long timeout_ticks, unsigned int* timeouts) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();  // 100 MHz
  while (atomicAdd(sig, 0ull) < value) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {
      if (timeouts) atomicAdd(timeouts, 1u);
      break;
    }
  }
}
// double -> float of a small vector (SyncBatchNorm: the per-channel sums travel between the ranks in double, tfpp_bn_bwd_apply_rows takes float rows)
__global__ void f64_to_f32_kernel(const double* __restrict__ x, float* __restrict__ y, long n, double scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (float)(x[i] * scale);
}
extern "C" int tfpp_f64_to_f32(const double* x, float* y, int64_t n, double scale, void* stream) {
  if ((n > 0 && (!x || !y)) || n < 0) return TFPP_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, scale);
  TFPP_CHECK_LAUNCH();
  return 0;
}
// Self-describing form (round 5): the signal does not COUNT passes, it carries the serial number of the pass that raised it.
// tfpp_set_u64 writes the host's serial of the pass about to be issued into a device word on the compute stream (outside any captured
// graph: stream-ordered in front of the replay); the in-graph node tfpp_signal_set raises sig to max(sig, *serial_word); the wait looks
// for sig >= the serial the host gave THIS pass.  A pass the host never book-kept (a bare graph.replay(), an aborted eager pass) re-raises
// an old serial: a later wait can be satisfied late (time-out, reported), never early on gradients still being written.
__global__ void set_u64_kernel(unsigned long long* p, unsigned long long v) { *p = v; }
__global__ void signal_set_kernel(unsigned long long* sig, const unsigned long long* serial) { atomicMax(sig, *serial); }
extern "C" int tfpp_set_u64(uint64_t* p, uint64_t v, void* stream) {
  if (!p) return TFPP_EINVAL;
  hipLaunchKernelGGL(set_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)p, (unsigned long long)v);
  TFPP_CHECK_LAUNCH();
  return 0;
}
extern "C" int tfpp_signal_set(uint64_t* sig, const uint64_t* serial, void* stream) {
  if (!sig || !serial) return TFPP_EINVAL;
  hipLaunchKernelGGL(signal_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)sig, (const unsigned long long*)serial);
  TFPP_CHECK_LAUNCH();
  return 0;
}
extern "C" int tfpp_signal_wait(uint64_t* sig, uint64_t value, int timeout_ms, uint32_t* timeouts, void* stream) {
  if (!sig || timeout_ms < 1) return TFPP_EINVAL;
  hipLaunchKernelGGL(signal_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)sig, (unsigned long long)value,
                     (unsigned long long)timeout_ms * 100000ull, (unsigned int*)timeouts);
  TFPP_CHECK_LAUNCH();
  return 0;
}
