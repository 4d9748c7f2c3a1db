/* tfpp.h -- C ABI of libtfpp_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the TransFuser++ hot path.
 *
 * The reference (autonomousvision/carla_garage) has no FFI/plugin layer: its seam is the Python class
 * team_code/model.py:24 `LidarCenterNet` whose forward (model.py:279-392) dispatches ~1100 ATen/cuDNN operator
 * calls.  Each entry point below replaces one class of those operator calls;
 * the citation on every function names the reference call site it stands in for.  The Python boundary module
 * carla_garage_amd/model.py is the only caller (ctypes binding in carla_garage_amd/_lib.py, shown in
 * INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator); the library never
 *     allocates, frees or synchronises; `stream` is a hipStream_t passed as void*; launches are asynchronous.
 *   - dtype: TFPP_F32 (0) or TFPP_BF16 (1) selects the activation/weight element type; accumulation is fp32.
 *   - activations are NHWC ([B,H,W,C] == tokens [B,T,C]); weights are pre-packed by tfpp_pack_* from the
 *     reference's state_dict layouts (OIHW / [out,in]).
 *   - return value: 0 on success, negative hipError_t on a launch error, TFPP_EINVAL (-1000) on bad arguments.
 */
#ifndef TFPP_H_
#define TFPP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TFPP_ABI_VERSION 8
#define TFPP_EINVAL (-1000)
#define TFPP_F32 0
#define TFPP_BF16 1
#define TFPP_ACT_NONE 0
#define TFPP_ACT_RELU 1
#define TFPP_ACT_SIGMOID 2
#define TFPP_ACT_GELU 3
#define TFPP_ACT_TANH 4

int tfpp_version(void);
/* 64-bit hash of the csrc/ + include/ sources the library was compiled from (0: built without it); checked at load. */
int tfpp_source_hash(uint64_t* out);
/* sizeof() of the parameter structs, in declaration order, so the ctypes mirror can be verified. */
int tfpp_struct_sizes(int* out, int n);

/* ---------------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d whose statistics still lie in the per-M-tile rows that the producing convolution's epilogue wrote
 * (tfpp_conv_params.stats_partial with stats_store = 1).  Round 6: there is no finalize launch any more on the layers whose row count is
 * small (<= TFPP_BN_ROWS_MAX) -- the FIRST kernel that consumes the normalised tensor adds the rows of the channels it touches in its
 * prologue (in double, fixed order: bit-reproducible), beside its own first loads, and one designated workgroup per channel block writes
 * scale / shift / the saved statistics and updates the running statistics, exactly what tfpp_bn_finalize_partials did.  Later kernels of
 * the pass (and the backward pass) pass partial = NULL and READ scale / shift.  (timm ConvNormAct, F.batch_norm in training mode.) */
#define TFPP_BN_ROWS_MAX 256
typedef struct {
  const float* partial;  /* [nrows][2*C] sums | sums of squares of the raw convolution output; NULL: scale / shift are final, read them */
  int nrows, C;
  int64_t count;         /* elements per channel behind the sums: B*H*W */
  const float* gamma; const float* beta;                                   /* nullable: 1 / 0 */
  float* running_mean; float* running_var; int64_t* num_batches_tracked;   /* nullable; touched only when partial != NULL */
  float* scale; float* shift;                                              /* [C] written when partial != NULL, read otherwise */
  float* save_mean; float* save_invstd;                                    /* nullable; written when partial != NULL */
  float momentum, eps;
} tfpp_bn_rows;

/* ---------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer (MFMA).  Replaces F.conv2d / nn.Linear calls:
 * timm RegNet 1x1 + grouped 3x3 convs (team_code/transfuser.py:25,52-55), fusion linears
 * (transfuser.py:352-359,392-394), token 1x1 convs (transfuser.py:85-94), FPN / heads / decoders
 * (transfuser.py:125-137, center_net.py:43-47, transfuser_utils.py:674-695, model.py:75-90,118-119,148).
 *   mode 0 (forward):  dst[b,hd,wd,g*n_g+n] = act(alpha*sum_{r,s,c} src[b,hd*stride-pad+r,wd*stride-pad+s,g*ks_g+c]
 *                                                  * w[g][n][(r*S+s)*ks_g+c] * scale[.] + shift[.] + res[...])
 *   mode 1 (data gradient): dst is the input-gradient [B,Hd,Wd,Cd], src the output-gradient [B,Hs,Ws,Cs]:
 *                      dst[b,h,w,g*n_g+n] = sum_{r,s,c} src[b,(h+pad-r)/stride,(w+pad-s)/stride,g*ks_g+c] * w[g][n][(r*S+s)*ks_g+c]
 *                      (terms with a non-integer or out-of-range source pixel are zero); `w` is the
 *                      tfpp_pack_conv_weight(..., transpose=1) image.
 * A linear layer is B=rows, H=W=1, R=S=1.  ks_g and K=R*S*ks_g must be multiples of 8 (bf16) / 4 (f32). */
typedef struct {
  const void* src; const void* w; void* dst;
  const float* scale;   /* per destination channel, nullable */
  const float* shift;   /* per destination channel (bias or folded BN shift), nullable */
  const void* res;      /* residual, same dtype, NHWC with pixel stride res_ld, nullable */
  int B, Hs, Ws, Cs;    /* source tensor */
  int Hd, Wd, Cd;       /* destination tensor */
  int R, S, stride, pad;
  int G, ks_g, n_g;     /* groups, source channels per group, destination channels per group */
  int mode, act, dst_nchw;
  float alpha;
  int64_t src_ld, dst_ld, res_ld; /* pixel strides in elements (allow channel-slice views) */
  int dst_f32;          /* 1: dst is float regardless of dtype */
  float* stats_partial; /* nullable: fused BatchNorm statistics, [stats_rows][2*Cd] fp32 partial sums / sums of squares, zeroed by the caller */
  int stats_rows;       /* M-tile t accumulates into row t % stats_rows */
  int stats_store;      /* 1 (requires stats_rows = tfpp_conv_gemm_stats_rows()): every (row, channel) cell has exactly one writer, so the
                           epilogue STORES instead of adding -- the rows need no zeroing and stay valid until the next launch writes them
                           (the consumers of tfpp_bn_rows.partial read them in their prologues) */
  float* splitk_ws;     /* nullable: fp32 workspace for split-K (few output tiles x long reduction: the fp32 planning head,
                           K = 9*1512 FPN convs); the dispatcher splits K over up to splitk_ws_floats / (M*Cd) workgroups per
                           tile and a second kernel sums the slices and applies the epilogue */
  int64_t splitk_ws_floats;
  int splitk;           /* must be 0 (set by the dispatcher on its own copy) */
  /* Fused BatchNorm-backward statistics (mode 1, bf16 NHWC destination, no split-K: see tfpp_conv_gemm_bns_ok).  When this call
   * produces the COMPLETE gradient g' of a tensor y = act(BN(x)) (the forward input of the convolution), the epilogue also emits,
   * per M-tile t and destination channel c, the sums the BatchNorm backward needs:
   *     g = g' * (y > 0 if bns_relu)        bns_partial[t][c] = sum g        bns_partial[t][Cd + c] = sum g * (x - mean[c]) * invstd[c]
   * (plain stores, every cell written exactly once: deterministic, nothing to zero) -- tfpp_bn_bwd_apply_rows finishes the job and
   * the separate read pass of tfpp_bn_bwd_reduce over g', y and x disappears.  g is taken after rounding to bf16, i.e. exactly the
   * values tfpp_bn_bwd_apply_rows reads back. */
  const void* bns_y;        /* nullable unless bns_relu: forward value of the tensor whose gradient is produced, [M][bns_ld] */
  const void* bns_x;        /* the BatchNorm input (raw convolution output), [M][bns_ld] */
  const float* bns_mean; const float* bns_invstd;
  float* bns_partial;       /* nullable = feature off; [tfpp_conv_gemm_stats_rows()][2*Cd] */
  int64_t bns_ld;
  int bns_relu;
  /* Round 6: the SOURCE is a tensor that exists only as (raw convolution output, BatchNorm statistics): src = raw and the loader applies
   * v = raw * scale[c] + shift[c], ReLU if in_relu, while it stages the tile (zero padding stays zero) -- the normalised tensor is never
   * written.  Only the register-staged 3x3 LDS-halo kernel can do this (tfpp_conv_gemm_in_bn_ok); in_bn.partial != NULL additionally runs the
   * finalize prologue described at tfpp_bn_rows.  in_bn.scale == NULL: feature off. */
  tfpp_bn_rows in_bn;
  int in_relu;
  /* Round 6 (data gradients): the tensor whose gradient this call completes went through a ReLU (conv + bias + ReLU layers without
   * BatchNorm: transfuser_utils.py:685-704 decoders, center_net.py heads).  relu_mask = that tensor's FORWARD value [M][relu_mask_ld]
   * (bf16, same row layout as dst): the finished result (residual / pending gradient added) is zeroed where the forward value is <= 0,
   * i.e. the activation's backward runs in this epilogue instead of as a pass of its own.  tfpp_conv_gemm_relu_mask_ok() says whether
   * the kernel the dispatcher picks can do it (vector epilogue, no K split). */
  const void* relu_mask;
  int64_t relu_mask_ld;
} tfpp_conv_params;
/* number of K slices the dispatcher would use for p (1 = no split) */
int tfpp_conv_gemm_splits(const tfpp_conv_params* p, int dtype);
int tfpp_conv_gemm(const tfpp_conv_params* p, int dtype, void* stream);
/* kernel variant the dispatcher picks for (p, dtype) -- used by the bench's per-kernel bookkeeping.  0..3: LDS-staged
 * 128x32 / 128x64 / 64x64 / 128x128 tiles; 200 / 201: LDS-DMA ring 128x128 / 64x128 (bf16, N >= 128, K >= 512); 300 + FN: 3x3 stride-1
 * LDS-halo kernel with FN 16-channel output fragments. */
int tfpp_conv_gemm_variant(const tfpp_conv_params* p, int dtype);
int tfpp_conv_gemm_mtiles(const tfpp_conv_params* p);
/* exact number of M-tiles of the kernel that runs for (p, dtype): with stats_rows = that, every (row, channel) cell of
 * stats_partial receives exactly one addend and the fused BatchNorm statistics are bit-reproducible */
int tfpp_conv_gemm_stats_rows(const tfpp_conv_params* p, int dtype);
/* 1 if the kernel the dispatcher runs for (p, dtype) supports the fused BatchNorm-backward statistics (bns_* fields) */
int tfpp_conv_gemm_bns_ok(const tfpp_conv_params* p, int dtype);
/* 1 if the kernel the dispatcher runs for (p, dtype) can normalise its source while loading it (in_bn) */
int tfpp_conv_gemm_relu_mask_ok(const tfpp_conv_params* p, int dtype);  /* 1: tfpp_conv_params.relu_mask is honoured for this launch */
int tfpp_conv_gemm_in_bn_ok(const tfpp_conv_params* p, int dtype);
/* debugging aid (TFPP_GLDS_TRACE=1): per-workgroup phase timestamps of the last LDS-DMA GEMM launch; returns slots per workgroup */
int tfpp_debug_glds_trace(uint64_t* out, int n_blocks);

/* Weight gradient of the same convolution (autograd of F.conv2d / F.linear, train.py:898):
 *   dw[(g*n_g+n), c, r, s] += sum_{b,hd,wd} dy[b,hd,wd,g*n_g+n] * x[b,hd*stride-pad+r,wd*stride-pad+s,g*ks_g+c]
 * accumulated into `dw` (reference layout OIHW, [Cout][c_real][R][S]).  The pixel reduction is split over `splits`
 * workgroups per output tile; with a workspace the slices are stored as fp32 tiles and summed by a second kernel
 * (deterministic; device-scope fp32 atomics measured 2-7x slower on gfx950), without one they are added with atomics.
 * Channels c >= c_real
 * (zero padding of the stem input) are dropped; row_map (nullable) maps packed output rows to parameter rows
 * (-1 = padding row) for the head-padded fused QKV weight. */
typedef struct {
  const void* dy; const void* x; float* dw;
  const int* row_map;   /* nullable: packed output row -> parameter row (-1 = padding row) */
  const int* col_map;   /* nullable: packed column kk -> parameter column (-1 = padding); default OIHW formula */
  int B, Hs, Ws, Cs, Hd, Wd, Cd;
  int R, S, stride, pad;
  int G, ks_g, n_g, c_real;
  int splits;           /* <=0: chosen by the library */
  int64_t x_ld, dy_ld;
  int64_t dw_ld;        /* elements per output row of dw (c_real*R*S unless rows are wider) */
  float* ws;            /* nullable: workspace for the pixel-split slices, [splits][G*n_g][R*S*ks_g] fp32 */
  int64_t ws_floats;
  /* Round 6: x exists only as (raw convolution output, final BatchNorm scale / shift): the loader applies x = raw * x_scale[c] + x_shift[c]
   * (ReLU if x_relu) while staging; 3x3 LDS-halo weight-gradient kernel only (tfpp_conv_wgrad_x_bn_ok).  x_scale == NULL: feature off. */
  const float* x_scale; const float* x_shift;
  int x_relu;
} tfpp_wgrad_params;
int tfpp_conv_wgrad(const tfpp_wgrad_params* p, int dtype, void* stream);
/* 1 if the kernel the dispatcher runs for (p, dtype) can normalise x while loading it (x_scale / x_shift / x_relu) */
int tfpp_conv_wgrad_x_bn_ok(const tfpp_wgrad_params* p, int dtype);
/* n independent weight gradients in one call (the batches of the training step's weight-gradient lane; the reference computes them one
 * autograd node at a time, train.py:898 loss.backward()).  Pointwise bf16 layers run as GROUPED grids -- the workgroups of up to 42 layers
 * in one launch, the descriptor table in the kernel arguments, pixel splits chosen for the group (most layers then add their tile straight
 * into dw; the rest share one slice-sum launch; no atomics) -- everything else through tfpp_conv_wgrad, in the order given.  All items use
 * items[i].ws as the slice workspace (the same buffer for every item of a call).  Results equal n tfpp_conv_wgrad calls up to the order of
 * the fp32 pixel sums.  TFPP_WGRAD_GROUP=0 disables grouping, TFPP_WGRAD_GROUP_WGS=n caps the grid (persistent workgroups). */
int tfpp_conv_wgrad_batch(const tfpp_wgrad_params* items, int n, int dtype, void* stream);
/* 1 if tfpp_conv_wgrad_batch runs this layer inside a grouped grid (the tile class -- 64 or 128 -- is the variant tfpp_conv_wgrad_stage plans) */
int tfpp_conv_wgrad_group_ok(const tfpp_wgrad_params* p, int dtype);
/* Preferred workspace of one call in bytes (the library never allocates): the size at which the dispatcher's plan is not
 * limited by the workspace.  op 0: split-K slices of tfpp_conv_gemm (params = tfpp_conv_params, field splitk_ws); op 1: pixel slices of
 * tfpp_conv_wgrad (params = tfpp_wgrad_params, field ws); op 2 / 3: BatchNorm / column-sum scratch for C = *(const int*)params channels.
 * A smaller workspace is legal everywhere: the kernels then use fewer slices (or atomics). */
int tfpp_workspace_bytes(int op, const void* params, int dtype, int64_t* bytes_out);

/* the same call in separately launchable pieces (per-kernel timing): stage 1 = first-stage kernel, 2 = slice sum, -1 = plan only;
 * plan_out[3] (nullable) = {variant: 0 LDS-staged 32x32, 1 LDS-staged 64x64, 2 LDS-DMA ring 64x64, 3 3x3 halo, 4 LDS-DMA ring 128x128
 * (8 waves); slices; has second stage} */
int tfpp_conv_wgrad_stage(const tfpp_wgrad_params* p, int dtype, int stage, int* plan_out, void* stream);

/* Strided batched GEMM C[z] = act(alpha * A[z] x B[z]^T-or-not + bias): attention products
 * (transfuser.py:372-375, nn.MultiheadAttention inside nn.TransformerDecoderLayer model.py:137-143) and their
 * gradients.  a_km=0: A[m*lda+k]; a_km=1: A[k*lda+m].  b_km=0: B[n*ldb+k]; b_km=1: B[k*ldb+n].
 * batch z in [0, batch0*batch1): offset = (z / batch1) * bs0 + (z % batch1) * bs1. */
typedef struct {
  const void* A; const void* B; void* C; const float* bias;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int64_t a_bs0, a_bs1, b_bs0, b_bs1, c_bs0, c_bs1;
  int batch0, batch1;
  int a_km, b_km, act, c_f32;
  float alpha;
  float beta;           /* C = result + beta*C_old (0: overwrite) */
} tfpp_bgemm_params;
int tfpp_bgemm(const tfpp_bgemm_params* p, int dtype, void* stream);
/* kernel the dispatcher picks for p: 0 = 64 x 64 tiles, 1 = the small-problem kernel (fp32, 32 x 32 tiles, K split over the four waves of a workgroup) */
int tfpp_bgemm_variant(const tfpp_bgemm_params* p, int dtype);

/* ---------------------------------------------------------------------------------------------------------
 * Fused multi-head self-attention of the fusion transformers (team_code/transfuser.py:362-380: q k^T / sqrt(d) -> softmax ->
 * attn_drop -> @ v), forward and backward, bf16.  q, k, v are head-major column slices of token matrices: element (b, t, h, e) at
 * base + (b*T + t) * ld + h*d + e; o / d_o likewise with ld_o.  T a multiple of 64, <= 320; d (storage head dim, zero-padded) a
 * multiple of 8, <= 384.  The scores stay in registers; the forward saves lse[b*nh*T] = log sum_j exp(scale * s_ij) and the
 * backward recomputes the probabilities from it.  Dropout uses the hash, seed and element index (row * T + key) of
 * tfpp_softmax_fwd, so fused and unfused paths draw the same masks.  delta: backward scratch [B*nh*T] floats.
 * debug_p (nullable): the forward also writes the dropped-out probabilities [B*nh*T][T] fp32 (tests). */
typedef struct {
  const void* q; const void* k; const void* v;
  void* o;              /* forward: output; backward: the saved forward output (read) */
  float* lse;           /* forward: written; backward: read */
  const void* d_o;      /* backward: gradient of o */
  void* dq; void* dk; void* dv; /* backward outputs, strides of q / k / v */
  float* delta;         /* backward scratch */
  float* debug_p;
  int B, nh, T, d;
  int64_t ld_q, ld_kv, ld_o;
  float scale, p_drop;
  uint64_t seed;
  const uint64_t* seed_offset;
  /* window attention (tfpp_attn_window_fwd, Video-Swin WindowAttention3D): dense additive terms and the saved probabilities */
  const float* bias;   /* [nh][T][ld_b]: relative-position bias of (head, query, key) */
  const float* mask;   /* [n_mask][T][ld_b]: shift mask of window b % n_mask, or NULL */
  void* p_out;         /* NULL, or bf16 [B][nh][T][ld_p]: softmax probabilities for the backward (columns >= T are left untouched) */
  int n_mask;
  int64_t ld_b, ld_p;
} tfpp_attn_params;
int tfpp_attn_supported(const tfpp_attn_params* p, int dtype); /* 1 if the fused kernels handle (p, dtype) */
int tfpp_attn_fwd(const tfpp_attn_params* p, int dtype, void* stream);
int tfpp_attn_bwd(const tfpp_attn_params* p, int dtype, void* stream);
/* The same forward for the 3-D shifted-window attention of the Video-Swin branch (team_code/video_swin_transformer.py:139-166): B = windows,
 * T = tokens per window (147: any T <= 320), softmax(scale * q k^T + bias[h] + mask[b % n_mask]) v; bias / mask are dense fp32 with row pitch
 * ld_b (a multiple of 4, >= T); no dropout.  One workgroup per (window, head, 64 queries): K, V and the scores of a window never leave the CU.
 * window_bias_dense expands relative_position_bias_table through relative_position_index into that dense layout. */
int tfpp_attn_window_fwd(const tfpp_attn_params* p, int dtype, void* stream);
int tfpp_window_bias_dense(const float* table, const int32_t* rel_index, float* dense, int heads, int n, int64_t ld_b, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Weight packing (state_dict layout -> kernel layout, also casts fp32 -> dtype).
 * tfpp_pack_conv_weight: OIHW [Cout][cin_g][R][S] ->
 *     transpose=0 (forward)       [G][n_pad][(r*S+s)*ks_pad + c]      (zero rows n>=n_g, zero channels c>=cin_g)
 *     transpose=1 (data gradient) [G][cin_g][(r*S+s)*n_pad + n]
 * tfpp_pack2d: out[r][c] = in[row_map[r]][col_map[c]] (transpose_in: in[col_map[c]][row_map[r]]); a -1 map entry
 *     yields 0; NULL maps are the identity.  Used for linear weights (head-padded fused QKV, transposes). */
int tfpp_pack_conv_weight(const float* w, void* out, int Cout, int cin_g, int R, int S, int G, int ks_pad, int n_pad, int transpose,
                          int dtype, void* stream);
int tfpp_pack2d(const float* in, void* out, const int* row_map, const int* col_map, int rows_out, int cols_out, int64_t in_ld,
                int64_t out_ld, int transpose_in, int dtype, void* stream);
int tfpp_cast(const void* in, void* out, int64_t n, int dtype_in, int dtype_out, void* stream);
/* Widening of the narrow host dtypes of an uploaded batch on the device (team_code/train.py:688-766 does ``.to(device, dtype=...)`` from pageable
 * memory): src_kind 0 = uint8, 1 = int32; dst_kind 0 = fp32, 1 = int64.  16-byte aligned buffers. */
int tfpp_widen(const void* in, void* out, int64_t n, int src_kind, int dst_kind, void* stream);

/* Colour augmentation of the uploaded uint8 camera frames (team_code/data.py:1141-1157 image_augmenter, applied per sample in
 * CARLA_Data.__getitem__, data.py:481-496; imgaug 0.4.0 / OpenCV 4.6, requirements.txt:48,95).  The HOST samples every image's program --
 * which of the seven operators fire (Sometimes(prob)), their order (Sequential(random_order=True)) and parameters -- into
 * progs[B][TFPP_AUG_MAX_OPS] (kind TFPP_AUG_NONE pads short programs; device copy); one call executes stage `stage` of every image:
 * dst = op(src) on (B, 3, H, W) uint8 planes (the loader's CHW RGB, data.py:516), src != dst.  Pixels are uint8 between stages as between
 * imgaug augmenters.  Per-pixel random maps (noise, dropout mask, displacement field) are drawn from a counter-based generator keyed by
 * (seed, image, stage, pixel, channel).  Operator parameters in tfpp_aug_op.a:
 *   BLUR      a[0..2] = 1-D Gaussian weights w(0), w(1), w(2) of the 5-tap kernel (imgaug: ksize 5 for sigma <= 1.5), BORDER_REFLECT_101
 *   NOISE     a[0] = sigma of the additive Gaussian noise;  per_channel: a noise map per channel instead of one shared map
 *   DROPOUT   a[0] = probability of zeroing a pixel;        per_channel: per channel value
 *   MULTIPLY  a[c] = factor (a[0] for all channels unless per_channel);   table clip(round(v m))
 *   CONTRAST  a[c] = alpha:  clip(127 + alpha (v - 127)) truncated to uint8
 *   GRAYSCALE a[0] = alpha:  round(alpha gray + (1 - alpha) v), gray = (4899 R + 9617 G + 1868 B + 8192) >> 14
 *   ELASTIC   a[0] = alpha, a[1..3] = smoothing weights w(0), w(1), w(2) of the U(-1, 1) displacement field; bicubic remap, border 0
 *   CUTOUT    a = x1, y1, x2, y2 (pixels), filled with the constant `per_channel` (128 for the camera, 0 for the LiDAR: data.py:1153,1165) */
enum { TFPP_AUG_NONE = 0, TFPP_AUG_BLUR = 1, TFPP_AUG_NOISE = 2, TFPP_AUG_DROPOUT = 3, TFPP_AUG_MULTIPLY = 4, TFPP_AUG_CONTRAST = 5,
       TFPP_AUG_GRAYSCALE = 6, TFPP_AUG_ELASTIC = 7, TFPP_AUG_CUTOUT = 8 };
#define TFPP_AUG_MAX_OPS 8
typedef struct tfpp_aug_op {
  int kind;
  int per_channel;
  float a[4];
} tfpp_aug_op;
int tfpp_image_augment_stage(const void* src, void* dst, const tfpp_aug_op* progs_dev, int stage, int B, int H, int W, uint64_t seed,
                             void* stream);
/* All per-step weight images in ONE launch: a device-resident table of descriptors (kind 0/1 = tfpp_pack_conv_weight
 * forward / transposed with a = {Cout, cin_g, R, S, G, ks_pad, n_pad}; kind 2 = tfpp_pack2d with a = {rows_out, cols_out,
 * transpose_in}).  Descriptor i owns workgroups [blk_start, blk_start + tfpp_pack_desc_plan(&desc_i)). */
typedef struct {
  const float* src; void* dst; const int* row_map; const int* col_map;
  int64_t total, in_ld, out_ld, blk_start;
  int kind, dtype;
  int a[8];
} tfpp_pack_desc;
int tfpp_pack_elems_per_block(void);
/* host side of the table: picks the packing path for *d (stored in d->a[7]: 0 element-wise, 1 contiguous cast, 2 LDS-tiled
 * transpose) and returns the number of workgroups the descriptor owns (blk_start of the next one = blk_start + that). */
int tfpp_pack_desc_plan(tfpp_pack_desc* d);
int tfpp_pack_multi(const tfpp_pack_desc* descs_dev, int n, int64_t total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * LiDAR point cloud -> BEV histogram, the producer of `lidar_bev`: replaces
 * CARLA_Data.lidar_to_histogram_features (team_code/data.py:873-906, called per tick at team_code/sensor_agent.py:421-425).
 * points: [n][point_stride >= 3] fp32 (x, y, z, ...); xedges/yedges: the nx+1 / ny+1 float64 bin edges of
 * np.linspace(min, max, (max-min)*int(pixels_per_meter)+1); counts: int32 scratch [C*ny*nx]; out: [C][ny][nx] fp32 with
 * C = 2 (below, above lidar_split_height) if use_ground_plane else 1 (above).  numpy.histogramdd bin semantics, counts
 * clipped at hist_max and divided by it (float64 divide, then float32).  Bit-exact with the reference. */
int tfpp_lidar_histogram(const float* points, int64_t n, int point_stride, const double* xedges, int nx, const double* yedges, int ny,
                         int32_t* counts, float* out, float max_height, float split_height, int use_ground_plane, int hist_max,
                         void* stream);

/* The loader's LiDAR path for a whole batch: CARLA_Data.align (team_code/data.py:840-871 = two
 * transfuser_utils.algin_lidar calls, transfuser_utils.py:116-130) + lidar_to_histogram_features (data.py:873-906) per frame, called per
 * sample and time frame from CARLA_Data.__getitem__ (data.py:524-560).  points: the raw float64 sweeps (laspy .xyz) of all frames
 * concatenated, (total_points, 3); offsets: int64 [frames + 1] (device); xforms: float64 [frames][10] = (t1, cos1, sin1, t2, cos2, sin2),
 * the two translations and the cos / sin of the two yaw angles as the host computes them (carla_garage_amd.lidar.align_params follows
 * data.py:853-868); out: float32 (frames, C, ny, nx); counts: int32 scratch of the same element count; aligned_out (nullable): the
 * aligned float64 points (tests).  float64 arithmetic in numpy's evaluation order: histograms equal the reference's bit for bit. */
int tfpp_lidar_align_histogram(const double* points, const int64_t* offsets, int64_t total_points, const double* xforms, int frames,
                               const double* xedges, int nx, const double* yedges, int ny, int32_t* counts, float* out, double max_height,
                               double split_height, int use_ground_plane, int hist_max, double* aligned_out, void* stream);

/* CenterNet heat-map decode, the step right after forward() when boxes are requested (replaces
 * LidarCenterNetHead.decode_heatmap, team_code/center_net.py:172-237, incl. gaussian_target.py:186-264): 3x3 local maxima of
 * heat (B, ncls, H, W), top-k in descending score order (equal scores: lower flat index first), gathered wh / offset /
 * argmax(yaw_class) + yaw_res -> out (B, k, 9) = x, y, w, h in image pixels, yaw, velocity = 0, brake = 0, class, score.
 * All maps are the caller-facing fp32 NCHW tensors.  ncls*H*W <= 32768. */
int tfpp_centernet_decode(const float* heat, const float* wh, const float* offset, const float* yaw_class, const float* yaw_res, float* out,
                          int B, int ncls, int H, int W, int k, int num_dir_bins, float width_ratio, float height_ratio, void* stream);

/* Rotated-box IoU + greedy NMS on the device (replaces transfuser_utils.py:409-450: shapely polygons + numpy loop; sensor_agent.py:491):
 * boxes [n][stride] fp32 rows (x, y, half width, half height, yaw [rad], ..., confidence at conf_idx), n <= 1024.  keep[0..*count) = row
 * indices of the kept boxes, most confident first (equal confidences: higher row first); iou_out (nullable, [n*n] float64): the IoU matrix. */
int tfpp_nms_rotated(const float* boxes, int n, int stride, int conf_idx, double iou_threshold, float min_conf, int32_t* keep, int32_t* count,
                     double* iou_out, void* stream);
/* rows with confidence <= min_conf never enter the suppression (model.py:451).  tfpp_bb_image_to_metric: the n x 9 rows of
 * tfpp_centernet_decode (image pixels) -> vehicle coordinates in metres (model.py:447-459 + transfuser_utils.py:388-406), same float32 results. */
int tfpp_bb_image_to_metric(const float* boxes, float* out, int n, float pixels_per_meter, float min_x, float min_y, void* stream);

/* CenterNet training targets from the box list, the rasterisation the loader workers do per sample (replaces
 * CARLA_Data.get_targets, team_code/data.py:697-790, incl. gaussian_target.py:11-61,166-187 and center_net.py:240-254).
 * boxes: (B, max_boxes, 8) float64 rows x, y, extent_x, extent_y, yaw, speed, brake, class in BEV image pixels exactly as
 * parse_bounding_boxes emits them (data.py:565-570, unpadded there: counts[b] rows are valid); outputs are the trainer's label tensors:
 * heat (B, ncls, H, W), wh / offset / pixel_weight (B, 2, H, W), yaw_res / velocity (B, 1, H, W) fp32, yaw_class / brake (B, H, W) int64,
 * avg_factor (B) = max(1, number of heat-map cells equal to 1).  Every output is fully written (no pre-zeroing).  Scalar targets are
 * bit-exact with the reference (float64 path); heat-map values within 2 ulp (float32 exp). */
int tfpp_centernet_targets(const double* boxes, const int32_t* counts, float* heat, float* wh, float* offset, int64_t* yaw_class,
                           float* yaw_res, float* velocity, int64_t* brake, float* pixel_weight, float* avg_factor, int B, int max_boxes,
                           int ncls, int H, int W, int num_dir_bins, double width_ratio, double height_ratio, double min_overlap,
                           void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Boundary layout changes.  nchw_to_nhwc_affine = normalize_imagenet (transfuser_utils.py:542-551) fused with
 * NCHW->NHWC and zero channel padding to `cpad`: out[b,h,w,c] = c<C ? in[b,c,h,w]*mul[c]+add[c] : 0 (mul/add nullable).
 * nhwc_to_nchw writes the caller-facing fp32 NCHW outputs (optionally through an activation, e.g. the depth sigmoid
 * model.py:379); nchw_to_nhwc_pad brings the caller's NCHW output-gradients back (padding channels zero). */
int tfpp_nchw_to_nhwc_affine(const float* in, void* out, const float* mul, const float* add, int B, int C, int H, int W, int cpad,
                             int dtype, void* stream);
/* the same from the camera frame as the caller holds it (sensor_agent.py:277-286 after cv2.imdecode; train.py:750 before .to(float32)):
 * uint8, hwc = 1: [B,H,W,C] / hwc = 0: [B,C,H,W]; swap = 1 reverses the channel order (BGR -> RGB):
 * out[b,h,w,c] = c<C ? float(in[.., swap ? C-1-c : c]) * mul[c] + add[c] : 0. */
int tfpp_u8_to_nhwc_affine(const uint8_t* in, void* out, const float* mul, const float* add, int B, int C, int H, int W, int cpad, int hwc,
                           int swap, int dtype, void* stream);
int tfpp_nhwc_to_nchw(const void* in, float* out, int B, int C, int H, int W, int64_t in_ld, int act, int dtype, void* stream);
int tfpp_nchw_to_nhwc_pad(const float* in, void* out, int B, int C, int H, int W, int64_t out_ld, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * BatchNorm2d (timm ConvNormAct in every RegNet block; F.batch_norm).  x is [rows, C] (NHWC flattened).
 * bn_stats: ws[0:C] = sum x, ws[C:2C] = sum x^2 in double (two-stage reduction through `scratch`, no atomics).
 * bn_finalize (train): batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale; saves
 *   mean/invstd; running stats update with `momentum` and the unbiased variance; num_batches_tracked += 1.
 * bn_fold (eval): scale/shift from the running statistics (then applied in the conv epilogue).
 * affine_act: y = act(x*gate[b,c]*scale[c] + shift[c] + res)   (each of gate/scale/shift/res nullable; gate is the
 *   squeeze-excite scale, [B,C] fp32, rows_per_batch rows per sample).
 * bn_bwd_reduce: g = dy*(y>0 if relu_mask); stage-1 partials of sum g, sum g*xhat into scratch (ws nullable: if given it
 *   also receives the two sums in double).
 * bn_bwd_apply: reads the partials left in scratch by bn_bwd_reduce for the same (rows, C, dtype):
 *   dx = gamma*invstd*(g - s0/rows - xhat*s1/rows); dgamma += s1; dbeta += s0; dres = g (nullable); ws (nullable) <- s0,s1.
 * bn_finalize_partials: bn_finalize straight from `nrows` accumulation rows [nrows][2C] (tfpp_conv_params.stats_partial);
 *   clear != 0 re-zeroes the rows, so a buffer that starts zeroed needs no memset between layers. */
/* scratch: tfpp_bn_scratch_floats(C) floats shared by the three reduction entry points (stage-1 partials + coefficients) */
int tfpp_bn_scratch_floats(int C);
int tfpp_bn_stats(const void* x, float* scratch, double* ws, int64_t rows, int C, int dtype, void* stream);
/* second stage on its own: ws[v] = sum_k partial[k][v], v < n2c (used with tfpp_conv_params.stats_partial) */
int tfpp_bn_reduce_final(const float* partial, double* ws, int nblk, int n2c, void* stream);
int tfpp_bn_finalize(const double* ws, const float* gamma, const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean, float* save_invstd, int64_t rows,
                     int C, float momentum, float eps, void* stream);
int tfpp_bn_finalize_partials(float* partial, int nrows, int clear, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                              float* save_invstd, int64_t rows, int C, float momentum, float eps, void* stream);
int tfpp_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* scale,
                 float* shift, int C, float eps, void* stream);
int tfpp_affine_act(const void* x, const float* scale, const float* shift, const void* res, const float* gate, void* y, int64_t rows,
                    int C, int64_t rows_per_batch, int act, int dtype, void* stream);
int tfpp_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* save_mean, const float* save_invstd, float* scratch,
                       double* ws, int64_t rows, int C, int relu_mask, int dtype, void* stream);
int tfpp_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                      const float* save_invstd, double* ws, float* scratch, void* dx, void* dres, float* dgamma, float* dbeta,
                      int64_t rows, int C, int relu_mask, int dtype, void* stream);
/* bn_bwd_apply with the stage-1 sums supplied by the caller: `partial` = nrows rows of [2*C] (sum g, sum g*xhat), written by the
 * fused epilogue of the producing kernel (tfpp_conv_params.bns_partial, tfpp_se_bwd_apply's bns variant); coef: 3*C floats scratch. */
int tfpp_bn_bwd_apply_rows(const void* dy, const void* y, const void* x, const float* gamma, const float* save_mean,
                           const float* save_invstd, float* partial, int nrows, float* coef, void* dx, void* dres, float* dgamma,
                           float* dbeta, int64_t rows, int C, int relu_mask, int dtype, void* stream);
/* Round 6 (csrc/bn_rows_kernels.hip): the BatchNorm passes with the finalize / coefficient step in the PROLOGUE of the pass that needs it.
 * Thread layout: a workgroup owns a block of ~64 channels (8-9 sixteen-byte chunks) x 28-32 row slots, so the rows of statistics it has to add
 * are 0.25-0.5 KB wide instead of 2*C floats.
 * bn_apply_rows:  t = x * scale[c] + shift[c];  relu_pre: t = max(t, 0);  gate (nullable, [rows / rows_per_batch][C]): t *= gate;
 *                 res (nullable): t += res;  relu_post: t = max(t, 0);  y = t.   bn->partial != NULL: statistics from the rows (see tfpp_bn_rows).
 *                 Replaces tfpp_bn_finalize_partials + tfpp_affine_act (+ the squeeze-excite gate pass).
 * bn_bwd_reduce_rows: g = dy * mask;  partial[t][c] = sum g, partial[t][C + c] = sum g * (x - mean[c]) * invstd[c] over the rows of row block t;
 *                 mask: 0 none, 1 (y > 0) from the forward output y, 2 (x * scale[c] + shift[c] > 0) recomputed from the raw tensor (y == NULL:
 *                 the normalised tensor was never written).  Returns rows in *nrows_out (<= TFPP_BN_ROWS_MAX).
 * bn_bwd_apply_rows2: dx = gamma*invstd*(g - s0/rows - xhat*s1/rows) with s0, s1 = the column sums of `partial` added in the prologue;
 *                 dgamma += s1, dbeta += s0 (designated workgroups); dres = g (nullable).  Replaces tfpp_bn_bwd_apply_rows' two launches. */
int tfpp_bn_apply_rows(const void* x, const tfpp_bn_rows* bn, const void* res, const float* gate, void* y, int64_t rows, int64_t rows_per_batch,
                       int relu_pre, int relu_post, int dtype, void* stream);
int tfpp_bn_bwd_rows_count(int64_t rows, int C, int dtype);
int tfpp_bn_bwd_reduce_rows(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* save_mean,
                            const float* save_invstd, float* partial, int64_t rows, int C, int mask, int dtype, void* stream);
int tfpp_bn_bwd_apply_rows2(const void* dy, const void* y, const void* x, const float* scale, const float* shift, const float* gamma,
                            const float* save_mean, const float* save_invstd, const float* partial, int nrows, void* dx, void* dres,
                            float* dgamma, float* dbeta, int64_t rows, int C, int mask, int dtype, void* stream);
/* Squeeze-excite around a conv2 output that exists only as (raw, BatchNorm statistics) -- a2 = relu(BN2(raw2)) is never written:
 * mean_hw_bn: pool[b][c] = mean over HW of relu(x*scale+shift) (bn->partial != NULL: finalize prologue); one launch (ticket per sample).
 * se_bwd_apply_bn: dx = dy*gate[b,c] + dpool[b,c]/HW (the complete gradient of a2), and rows [2*C] of (sum g, sum g*xhat), g = dx (rounded)
 *   * (x*scale+shift > 0), for tfpp_bn_bwd_apply_rows2; rows = tfpp_se_bwd_apply_bn_rows(B, HW, C, dtype). */
int tfpp_mean_hw_bn(const void* x, const tfpp_bn_rows* bn, float* out, float* scratch, float* ticket_scratch, int B, int HW, int dtype, void* stream);
int tfpp_se_bwd_apply_bn_rows(int B, int HW, int C, int dtype);
int tfpp_se_bwd_apply_bn(const void* dy, const float* gate, const float* dpool, const void* x, const float* scale, const float* shift,
                         const float* save_mean, const float* save_invstd, void* dx, float* partial, int B, int HW, int C, int dtype, void* stream);
/* BatchNorm1d(1, affine=False) on the ego speed (model.py:216,311), fp32 [B]. */
int tfpp_bn1d_scalar(const float* x, float* y, float* running_mean, float* running_var, int64_t* nbt, int B, int training,
                     float momentum, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Squeeze-excite (timm SEModule in the RegNet-Y bottleneck): pool = mean over HW -> fc1 -> ReLU -> fc2 -> sigmoid.
 * mean_hw: [B,HW,C] -> [B,C] fp32.  se_gate_fwd: hidden [B,RD], gate [B,C].  The gate multiply is fused into
 * tfpp_affine_act.  Backward: se_dgate[b,c] = sum_hw dy*x; se_gate_bwd -> dpool + parameter gradients (atomics);
 * se_bwd_apply: dx = dy*gate[b,c] + dpool[b,c]/HW. */
/* scratch (mean_hw, se_dgate, colsum): tfpp_reduce_scratch_floats(B, C) floats of stage-1 partials; the column reductions are
 * two-stage (plain stores + a small second kernel), deterministic, no atomics or memsets. */
int tfpp_reduce_scratch_floats(int B, int C);
int tfpp_mean_hw(const void* x, float* out, float* scratch, int B, int HW, int C, int dtype, void* stream);
/* tfpp_mean_hw / tfpp_se_dgate in ONE launch (round 5): the row-block workgroups of a sample publish their partial sums, the workgroup that draws the
 * sample's last ticket adds them in a fixed order and writes the result.  ticket_scratch: a tfpp_gridsum_scratch_floats() buffer (zero before the
 * first use, left at zero, not shared between concurrent streams); B <= 64, else TFPP_EINVAL. */
int tfpp_mean_hw_ticket(const void* x, float* out, float* scratch, float* ticket_scratch, int B, int HW, int C, int dtype, void* stream);
int tfpp_se_dgate_ticket(const void* dy, const void* x, float* dgate, float* scratch, float* ticket_scratch, int B, int HW, int C, int dtype, void* stream);
int tfpp_se_gate_fwd(const float* pool, const float* w1, const float* b1, const float* w2, const float* b2, float* hidden,
                     float* gate, int B, int C, int RD, void* stream);
int tfpp_se_dgate(const void* dy, const void* x, float* dgate, float* scratch, int B, int HW, int C, int dtype, void* stream);
int tfpp_se_gate_bwd(const float* dgate, const float* gate, const float* hidden, const float* pool, const float* w1,
                     const float* w2, float* dz1_scratch /* [B*RD] */, float* dpool, float* dw1, float* db1, float* dw2, float* db2,
                     int B, int C, int RD, void* stream);
/* se_gate_bwd with dgate_g[b,c] = dgate[b,c] * gate[b,c] = sum_hw dy * y, y = the GATED tensor as the forward pass stored it (tfpp_se_dgate on
 * (dy, y)).  Round 6: the gate's gradient is a small residual of cancelling sums -- BatchNorm behind conv3 removes any common scale of its input --
 * and the cancellation holds for the tensor conv3 actually read, roundings included; rebuilding the un-gated activation instead
 * (tfpp_se_dgate_bn) left the squeeze-excite fc1 gradients of the bf16 step 1.4x further from the fp32 step's. */
int tfpp_se_gate_bwd_premul(const float* dgate_g, const float* gate, const float* hidden, const float* pool, const float* w1,
                            const float* w2, float* dz1_scratch, float* dpool, float* dw1, float* db1, float* dw2, float* db2, int B,
                            int C, int RD, void* stream);
int tfpp_se_bwd_apply(const void* dy, const float* gate, const float* dpool, void* dx, int B, int HW, int C, int dtype, void* stream);
/* se_bwd_apply with the BatchNorm-backward statistics of the preceding layer fused in (conv2 of a RegNet bottleneck: dx is the
 * complete gradient of y = relu(BN(x))): also writes tfpp_se_bwd_apply_bns_rows(B, HW, C, dtype) rows [2*C] of (sum g, sum g*xhat),
 * g = dx * (y > 0), for tfpp_bn_bwd_apply_rows. */
int tfpp_se_bwd_apply_bns_rows(int B, int HW, int C, int dtype);
int tfpp_se_bwd_apply_bns(const void* dy, const float* gate, const float* dpool, const void* y, const void* x, const float* save_mean,
                          const float* save_invstd, void* dx, float* partial, int B, int HW, int C, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Pooling / resampling: F.adaptive_avg_pool2d with uniform windows (transfuser.py:230-231) and F.interpolate
 * bilinear align_corners=False (transfuser.py:239-255,119-123; transfuser_utils.py:699-701; model.py:88-90).
 * bilinear_fwd: y = base + bilinear(x)*mul[pixel] (base, mul nullable); output NHWC (pixel stride y_ld) or, with
 *   out_nchw_f32, caller-facing fp32 NCHW with c_real channels.  bilinear_bwd is the exact adjoint in gather form. */
int tfpp_avgpool_fwd(const void* x, void* y, int B, int H, int W, int C, int Ho, int Wo, int64_t y_ld, int dtype, void* stream);
int tfpp_avgpool_bwd_add(const void* dy, void* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t dy_ld, int dtype, void* stream);
int tfpp_bilinear_fwd(const void* x, const void* base, const float* mul, void* y, int B, int Hi, int Wi, int Ho, int Wo, int C,
                      int64_t x_ld, int64_t y_ld, int out_nchw_f32, int c_real, int dtype, void* stream);
int tfpp_bilinear_bwd(const void* dy, const float* mul, void* dx, int B, int Hi, int Wi, int Ho, int Wo, int C, int64_t dy_ld,
                      int64_t dx_ld, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * bev_encoder backbone (team_code/bev_encoder.py:180-201): camera -> BEV lift.  out[b][i][j][c] = scale[i][j] * sum_z bilinear(feat[b], coords[j][i][z])
 * with feat (B, Hf, Wf, C) NHWC, coords (D, W, Z, 2) sample positions in feature pixels (grid_sample's align_corners=False un-normalisation
 * of create_projection_grid's output, zeros padding), scale (W, D) = valid_bev_pixels / bev_projection_normalizer in the transposed (image)
 * orientation; out (B, W, D, C).  bwd: dfeat (fp32, caller-zeroed) += adjoint, fp32 atomics. */
int tfpp_bev_lift_fwd(const void* feat, const float* coords, const float* scale, void* out, int B, int Hf, int Wf, int C, int D, int W,
                      int Z, int dtype, void* stream);
int tfpp_bev_lift_bwd(const void* dout, const float* coords, const float* scale, float* dfeat, int B, int Hf, int Wf, int C, int D, int W,
                      int Z, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Video-Swin LiDAR backbone (BASELINE config 5; team_code/video_swin_transformer.py, consumed at transfuser.py:44-50,151-155).
 * patchify3d: im2col of PatchEmbed3D's Conv3d(1, 96, kernel = stride = (2, 4, 4)) (:427-467): x fp32 (B, T, H, W) ->
 *   rows (b, t/2, h/4, w/4) x 32 values ordered (kt, kh, kw); the projection itself is tfpp_conv_gemm with K = 32.
 * gather_rows: dst[r][0..C) = (idx[r] >= 0 ? src[idx[r]][0..C) : 0) + (add ? add[r][0..C) : 0).  With index tables built once per
 *   stage on the host this is F.pad + torch.roll + window_partition of SwinTransformerBlock3D.forward_part1 (:233-249), its inverse
 *   window_reverse + roll + crop fused with the shortcut add (:250-260,276), and the four strided slices + cat of PatchMerging (:305-309).
 * softmax_window_bias: WindowAttention3D.forward (:146-163), in place on the scores of `windows` x `heads` matrices of n x n (row pitch
 *   ld): softmax_j(alpha * s + table[rel_index[i][j]][h] + mask[w % n_mask][i][j]); table = relative_position_bias_table (fp32),
 *   rel_index = the [:n, :n] corner of relative_position_index as int32, mask = compute_mask's 0 / -100 matrices or NULL.  n <= 256. */
int tfpp_patchify3d(const float* x, void* out, int B, int T, int H, int W, int dtype, void* stream);
int tfpp_gather_rows(const void* src, const int32_t* idx, const void* add, void* dst, int64_t rows, int C, int64_t src_ld,
                     int64_t dst_ld, int64_t add_ld, int dtype, void* stream);
int tfpp_softmax_window_bias(void* s, const float* table, const int32_t* rel_index, const float* mask, int64_t windows, int heads,
                             int n, int64_t ld, int n_mask, float alpha, int dtype, void* stream);
/* Training of the same branch.  drop_path: timm DropPath as SwinTransformerBlock3D uses it (:216,276-281): y = x * keep_b / (1 - p) with one
 *   Bernoulli draw per SAMPLE (samples x elems_per_sample elements), from the dropout hash of tfpp_softmax_fwd on (seed, b); the backward is the
 *   same call on the gradient.  window_bias_grad: dtable[t][h] += scale * sum over the pairs (i, j) with rel_index[i][j] = t of sum_w ds[w][h][i][j]
 *   (inv_ptr [ntab + 1] / inv_pairs [n * n]: the pairs i * n + j of every table entry, ascending: a dense window sum + a fixed-order gather, no atomics), the gradient of
 *   relative_position_bias_table from the score gradient tfpp_softmax_bwd produced (scale = 1 / alpha undoes its alpha).  The gathers are their
 *   own adjoints with the inverse index table (gather_rows with rev / fwd swapped). */
int tfpp_drop_path(const void* x, void* y, int64_t samples, int64_t elems_per_sample, float p, uint64_t seed, const uint64_t* seed_offset,
                   int dtype, void* stream);
int tfpp_window_bias_grad(const void* ds, const int32_t* inv_ptr, const int32_t* inv_pairs, int ntab, float* dense_scratch /* [heads*n*n] */,
                          float* dtable, int64_t windows, int heads, int n, int64_t ld, float scale, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Token ops: LayerNorm (transfuser.py:388-389,288; nn.TransformerDecoderLayer norms), row softmax with the
 * 1/sqrt(d) scale and attention dropout (transfuser.py:372-374), residual add + dropout (transfuser.py:399-400),
 * positional-embedding add (transfuser.py:325; model.py:302,318), activation gradients, bias gradients.
 * softmax_fwd: P = softmax(alpha*x) in place; pd (nullable) = dropout(P).  softmax_bwd (in place on dp):
 *   dP = dPd*mask; dS = alpha * P .* (dP - sum_j dP_j P_j).  Dropout masks are regenerated from (seed, index). */
int tfpp_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int64_t rows, int C,
                       float eps, int dtype, void* stream);
/* Grid-wide sums of the kernels below (LayerNorm parameter gradients, loss sums, the cross-entropy normaliser) are added in a FIXED order by
 * the workgroup that finishes last -- no fp32 atomics, so a training step is bit-reproducible -- through `scratch`:
 * tfpp_gridsum_scratch_floats() floats that are ZERO before the first use (the kernels leave the ticket counters at zero) and are not shared
 * by launches that may run concurrently (one buffer per stream). */
int tfpp_gridsum_scratch_floats(void);
/* layernorm_bwd: dgamma / dbeta nullable (dx only: scratch may then be NULL too) */
int tfpp_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                       float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, int dtype, void* stream);
/* layernorm_param_grad: the dgamma / dbeta part of layernorm_bwd alone (the engine runs it on the weight-gradient lane).
 * add_layernorm_fwd: sum = a + dropout(b) and y = LayerNorm(sum) in one launch -- the post-norm residual step of
 *   nn.TransformerDecoderLayer (model.py:137-143); dropout mask of tfpp_add_dropout (seed, flat element index).
 * add_layernorm_bwd: d_sum = LayerNorm backward (= gradient of a) and d_b = d_sum * the same mask, one launch. */
int tfpp_layernorm_param_grad(const void* dy, const void* x, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* scratch,
                              int64_t rows, int C, int dtype, void* stream);
int tfpp_add_layernorm_fwd(const void* a, const void* b, void* sum, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                           int64_t rows, int C, float eps, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype, void* stream);
int tfpp_add_layernorm_bwd(const void* dy, const void* sum, const float* gamma, const float* mean, const float* rstd, void* d_sum, void* d_b,
                           float* dgamma, float* dbeta, float* scratch, int64_t rows, int C, float p_drop, uint64_t seed,
                           const uint64_t* seed_offset, int dtype, void* stream);
int tfpp_softmax_fwd(void* x, void* pd, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype,
                     void* stream);
int tfpp_softmax_bwd(const void* p, void* dp_inout, int64_t rows, int cols, int64_t ld, float alpha, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                     int dtype, void* stream);
int tfpp_add_dropout(const void* a, const void* b, void* y, int64_t n, float p_drop, uint64_t seed, const uint64_t* seed_offset, int dtype, void* stream);
/* dropout masks are a pure function of (seed + *seed_offset, element index); seed_offset (nullable) is a device counter that
 * tfpp_inc_u64 advances once per training step so that a replayed hipGraph draws fresh masks. */
int tfpp_inc_u64(uint64_t* p, void* stream);
/* Completion signal from a kernel node inside a captured hipGraph to a stream outside of it (data-parallel gradient exchange,
 * team_code/train.py:516-520: the all-reduce of a gradient bucket starts while backward is still running).  sig: a zero-initialised 64-bit
 * word in device memory, raised by tfpp_signal_set (below) behind everything issued so far on its stream.
 * tfpp_signal_wait: the work issued on `stream` after this call starts once *sig >= value; gives up after timeout_ms and then adds 1 to
 * *timeouts (nullable). */
/* y[i] = (float)(x[i] * scale): the per-channel BatchNorm sums of the SyncBatchNorm path (train.py:511-512) are all-reduced in double and handed
 * to tfpp_bn_bwd_apply_rows as one float row (scale = 1 / world: that entry point normalises by the LOCAL row count it also walks). */
int tfpp_f64_to_f32(const double* x, float* y, int64_t n, double scale, void* stream);
/* Self-describing signals (round 5): tfpp_set_u64 stores v into *p on `stream` (the host's serial number of the pass it is about to issue,
 * written in front of the pass / the graph replay); tfpp_signal_set raises *sig to max(*sig, *serial) behind everything issued so far on
 * `stream` (a kernel node inside the captured step); the exchange then waits with tfpp_signal_wait(sig, serial of that pass).  A pass the
 * host did not book-keep re-raises an old serial and can never satisfy a later wait early. */
int tfpp_set_u64(uint64_t* p, uint64_t v, void* stream);
int tfpp_signal_set(uint64_t* sig, const uint64_t* serial, void* stream);
int tfpp_signal_wait(uint64_t* sig, uint64_t value, int timeout_ms, uint32_t* timeouts, void* stream);
int tfpp_add_bcast(const void* x, const float* bcast, void* y, int64_t n, int64_t period, int dtype, void* stream);
int tfpp_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, int dtype, void* stream);
int tfpp_axpy(const void* x, void* y, int64_t n, float a, int dtype, void* stream);
int tfpp_mul_pixmask(const void* x, const float* m, void* y, int64_t n, int64_t ld, int64_t HW, int dtype, void* stream);
/* out[c] += sum_rows x[row*ld + c]; scratch: tfpp_reduce_scratch_floats(1, C) floats (NULL -> slower atomic path) */
int tfpp_colsum(const void* x, float* out, float* scratch, int64_t rows, int C, int64_t ld, int dtype, void* stream);
int tfpp_sum_f32(const float* x, float* out, int64_t n, void* stream);
/* token plumbing (torch.cat / slicing / .repeat in transfuser.py:323,329-337 and model.py:318-324,352-355):
 * dst[b*dst_bs + dst_off + i] (+)= src[b*src_bs + src_off + i] for i in [0,n), with dtype conversion. */
int tfpp_copy_rows(const void* src, void* dst, int B, int64_t n, int64_t src_bs, int64_t src_off, int64_t dst_bs, int64_t dst_off,
                   int accumulate, int dtype_in, int dtype_out, void* stream);
int tfpp_zero(void* p, int64_t bytes, void* stream); /* hipMemsetAsync(p, 0, bytes) */
int tfpp_fill_bytes(void* p, int value, int64_t bytes, void* stream); /* hipMemsetAsync(p, value, bytes): debug poisoning (TFPP_DEBUG_POISON) */
/* debugging aid (tools/replay_bisect.py, TFPP_DEBUG_NODE_HASH): *slot += an order-independent 64-bit hash of the 32-bit words of
 * p[0, bytes) (bytes a multiple of 4; integer adds only, so equal bytes always give equal sums whatever the block order). */
int tfpp_hash_words(const void* p, int64_t bytes, uint64_t* slot, void* stream);
/* debugging aid (tools/lane_timeline.py): *slot = wall_clock64() (100 MHz) at the time this launch runs on its stream. */
int tfpp_stamp(uint64_t* slot, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Attention core of the fp32 planning decoder (nn.MultiheadAttention inside nn.TransformerDecoderLayer, model.py:137-146,352: 11 / 8
 * queries against themselves or 65 memory tokens, 8 heads of 32): softmax(scale * q k^T) -> dropout -> @ v as ONE launch, one workgroup
 * per (sample, head), instead of batched GEMM -> softmax -> batched GEMM (and ONE launch instead of five in backward).  tq <= 16,
 * tk <= 96, d <= 32 (tfpp_small_attn_supported).  q / k / v / o and their gradients: element (b, t, h, e) at base + (b*T + t)*ld + h*d + e.
 * p_save [B*nh*tq][tk]: the probabilities before dropout, written by forward, read by backward.  Dropout masks: hash, seed and element
 * index of tfpp_softmax_fwd (row * tk + key), so both paths draw the same masks. */
int tfpp_small_attn_supported(int tq, int tk, int d);
int tfpp_small_attn_fwd(const float* q, const float* k, const float* v, float* o, float* p_save, int B, int nh, int tq, int tk, int d,
                        int64_t ld_q, int64_t ld_kv, int64_t ld_o, float scale, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                        void* stream);
int tfpp_small_attn_bwd(const float* q, const float* k, const float* v, const float* p_save, const float* d_o, float* dq, float* dk,
                        float* dv, int B, int nh, int tq, int tk, int d, int64_t ld_q, int64_t ld_kv, int64_t ld_o, float scale,
                        float p_drop, uint64_t seed, const uint64_t* seed_offset, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * GRU waypoint / checkpoint decoder (model.py:857-867): h0 = enc(target_point); nn.GRU(256->64) over T steps;
 * Linear 64->2; cumsum over time.  gi = x W_ih^T + b_ih comes from tfpp_conv_gemm; these kernels run the
 * recurrence and its BPTT, fp32.  save: [B,T,4,H] = (r,z,n,h). */
int tfpp_gru_fwd(const float* gi, const float* h0, const float* w_hh, const float* b_hh, const float* w_dec, const float* b_dec,
                 float* save, float* out, int B, int T, int H, void* stream);
/* Backward: one workgroup per sample writes dgi / dh0 and ONE partial image of the parameter gradients into partial[b]
 * (tfpp_gru_bwd_partial_floats(B, H) floats in all; no atomics); tfpp_gru_bwd_reduce adds the B images to the gradient
 * destinations in sample order.  With dw_hh .. db_dec given tfpp_gru_bwd launches the reduce itself on the same stream; with all
 * four null the caller runs it later (the engine: on the weight-gradient lane). */
int tfpp_gru_bwd(const float* dout, const float* save, const float* h0, const float* w_hh, const float* b_hh, const float* w_dec,
                 float* dgi, float* dh0, float* partial, float* dw_hh, float* db_hh, float* dw_dec, float* db_dec, int B, int T, int H,
                 void* stream);
int tfpp_gru_bwd_reduce(const float* partial, float* dw_hh, float* db_hh, float* dw_dec, float* db_dec, int B, int H, void* stream);
int tfpp_gru_bwd_partial_floats(int B, int H); /* returns the count (not an error code) */

/* ---------------------------------------------------------------------------------------------------------
 * Losses (model.py:394-445, center_net.py:77-123, transfuser_utils.py:341-364).  Each call adds the (unweighted)
 * loss to *loss_out and writes d(weight*loss)/d(pred) into dpred (nullable) in one pass.  pred is [rows, ld] NHWC
 * with C real channels (padding channels get zero gradient); labels keep the reference's layouts.
 * ce_loss: class-weighted cross entropy, mean over non-ignored rows (label -1, or vis_mask[pix]==0: the BEV
 *   visibility trick model.py:427-429); with pix_weight the per-row loss is multiplied by
 *   pix_weight[(row/HW)*pw_bstride + row%HW] and divided by (*denom + denom_eps) instead (center_net.py:109).
 * ws: 2 floats (the normaliser sum_rows w[y] of the unweighted-pixel form); scratch: tfpp_gridsum_scratch_floats(), see tfpp_layernorm_bwd.
 * reg_loss: kind 0 L1, 1 smooth-L1, 2 gaussian focal; logical element (b,c,pix): pred[(b*HW+pix)*ld+c],
 *   target[(b*C+c)*HW+pix], elem_weight[(b*wC+(w_bcast?0:c))*HW+pix]; denominator (*denom+denom_eps)*denom_mul or B*C*HW. */
int tfpp_ce_loss(const void* pred, const int64_t* label, const float* class_weight, const float* vis_mask, const float* pix_weight,
                 int64_t pw_bstride, int64_t HW, const float* denom, float denom_eps, float weight, float* loss_out, void* dpred,
                 float* ws, float* scratch, int64_t rows, int C, int ld, float label_smoothing, float focal_gamma, int dtype, void* stream);
/* focal_gamma < 0: cross entropy as above.  focal_gamma >= 0: the focal loss of team_code/focal_loss.py:35-103 (config.use_focal_loss,
 * model.py:255-256): mean over ALL rows of class_weight[y] (1 - p_y)^gamma (-log p_y); not with pix_weight / vis_mask / label smoothing.
 * label_smoothing a in [0, 1) (nn.CrossEntropyLoss(weight, label_smoothing=a), model.py:252-265; not with pix_weight): per row
 * (1 - a) w[y] nll(y) + a / C sum_c w[c] nll(c), normalised by sum_rows w[y]. */
int tfpp_reg_loss(const void* pred, const float* target, const float* elem_weight, int wC, int w_bcast, const float* denom,
                  float denom_eps, float denom_mul, float weight, float* loss_out, void* dpred, float* scratch, int B, int C, int64_t HW,
                  int64_t ld, int kind, int dtype, void* stream);

/* config.multi_wp_output (model.py:151-163,326-331,401-411; train.py:440-441): two waypoint hypotheses and a path-selection logit.
 * min_l1_pair_loss: pair [B, 2, n] fp32 (both hypotheses of a sample side by side), label [B, n]; *loss_out += mean_b min_h mean_e |pair - label|,
 *   dpair (nullable) = d(weight * loss) / d pair (zero for the hypothesis that lost), sel_label[b] = 0 / 1 = the hypothesis that won (B <= 1024).
 * bce_logits_loss: nn.BCEWithLogitsLoss() of logit[b * ld] against y[b]; *loss_out += the mean over b; dlogit [B, ld] (nullable). */
int tfpp_min_l1_pair_loss(const float* pair, const float* label, float weight, float* loss_out, float* dpair, float* sel_label, int B, int n,
                          void* stream);
int tfpp_bce_logits_loss(const float* logit, int ld, const float* y, float weight, float* loss_out, float* dlogit, int B, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimizer: torch.optim.AdamW(amsgrad=True) (train.py:529-531) over a flat fp32 arena; g is scaled by grad_scale
 * first (1/world_size after a sum all-reduce). */
int tfpp_adamw_amsgrad(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, float grad_scale, void* stream);
/* The same with the two parameter groups of create_optimizer_groups (model.py:556-632, train.py:522-523 use_optim_groups): no_decay_bits holds
 * one bit per 4 consecutive elements of the arena starting at p (every parameter starts on a multiple of 4); set = weight_decay 0. */
int tfpp_adamw_amsgrad_groups(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float grad_scale, const uint32_t* no_decay_bits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TFPP_H_ */
