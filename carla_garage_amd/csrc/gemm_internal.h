// Internal (non-ABI) declarations shared between the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tfpp.h"

// multi-stage LDS-DMA implicit GEMM (gemm_glds.hip), bf16, N >= 128; variant codes 200 (128x128) / 201 (64x128) / 202 (256x128)
bool conv_glds_supported(const tfpp_conv_params& p, int dtype);
int conv_glds_variant(const tfpp_conv_params& p);
int conv_glds_bm(int variant);  // rows per M-tile
int conv_gemm_glds(const tfpp_conv_params& p, hipStream_t st);

// 3x3 / stride 1 / pad 1 with the input tile staged once in LDS (conv3x3_halo.hip), bf16, n_g <= 64; variant codes 300 + FN
bool conv_halo_supported(const tfpp_conv_params& p, int dtype);
bool conv_halo_in_bn_ok(const tfpp_conv_params& p, int dtype);  // the source may be normalised while it is staged (tfpp_conv_params.in_bn)
int conv_halo_variant(const tfpp_conv_params& p);
int conv_halo_mtiles(const tfpp_conv_params& p);
int conv_gemm_halo(const tfpp_conv_params& p, hipStream_t st);

// weight gradient with the LDS-DMA ring + hardware transpose reads (gemm_wgrad_glds.hip), bf16; tile = 64 (4 waves) or 128 (8 waves)
bool wgrad_glds_supported(const tfpp_wgrad_params& p, int dtype);
bool wgrad_glds128_preferred(const tfpp_wgrad_params& p);
int conv_wgrad_glds(const tfpp_wgrad_params& p, int tile, hipStream_t st);

// 3x3 / stride 1 weight gradient with LDS-staged halo tiles (wgrad3x3_halo.hip), bf16, n_g, ks_g <= 64: slices per group (0 = n/a)
int wgrad_halo_slices(const tfpp_wgrad_params& p, int dtype);
int conv_wgrad_halo(const tfpp_wgrad_params& p, int nblk, hipStream_t st);

// Grouped pointwise weight gradients (gemm_wgrad_glds.hip conv_wgrad_glds_group_kernel, gemm_kernels.hip wgrad_reduce_group_kernel): the
// descriptor table of one launch, passed BY VALUE as the kernel argument (<= 4 KB), so that a captured hipGraph node carries it and no
// device-side table has to be kept consistent with the replays.  One item = one layer: dW[n_g][KK] += dY[P][n_g]^T X[P][KK].
struct tfpp_wgrad_item {
  const void* dy; const void* x; float* dw; float* ws; const int* row_map; const int* col_map;
  int P, n_g, KK, c_real, splits, dy_ld, x_ld, dw_ld;
  int wg_start, wgs;  // workgroup range [wg_start, wg_start + wgs) of the launch (wg_start: a multiple of 8 = whole XCD rounds)
  int pin, pad_;      // >= 0: slices are units pinned to XCDs, unit u on XCD (pin + u) % 8 (gemm_wgrad_glds.hip); -1: the tile orders of the single-layer kernel
};
#define TFPP_WGRAD_GROUP_MAX 42
struct tfpp_wgrad_group {
  int n, total;  // items, workgroups (padded ranges included)
  tfpp_wgrad_item it[TFPP_WGRAD_GROUP_MAX];
};
static_assert(sizeof(tfpp_wgrad_group) <= 4096, "kernel arguments are limited to 4 KB");
#if defined(__HIPCC__)
__device__ __forceinline__ tfpp_wgrad_params tfpp_wgrad_item_params(const tfpp_wgrad_item& it) {
  tfpp_wgrad_params p;
  p.dy = it.dy; p.x = it.x; p.dw = it.dw; p.row_map = it.row_map; p.col_map = it.col_map;
  p.B = it.P; p.Hs = 1; p.Ws = 1; p.Cs = it.KK; p.Hd = 1; p.Wd = 1; p.Cd = it.n_g; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
  p.G = 1; p.ks_g = it.KK; p.n_g = it.n_g; p.c_real = it.c_real; p.splits = it.splits;
  p.x_ld = it.x_ld; p.dy_ld = it.dy_ld; p.dw_ld = it.dw_ld; p.ws = it.ws; p.ws_floats = 0;
  p.x_scale = nullptr; p.x_shift = nullptr; p.x_relu = 0;
  return p;
}
#endif
// host side of the grouped launch: items are pointwise bf16 layers the LDS-DMA kernel supports (wgrad_glds_group_ok); tile = 64 | 128
bool wgrad_glds_group_ok(const tfpp_wgrad_params& p, int dtype);
int conv_wgrad_glds_group(const tfpp_wgrad_group& grp, int tile, int grid_cap, hipStream_t st);
