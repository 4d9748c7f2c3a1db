// Internal (non-ABI) declarations shared between the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tfpp.h"

// multi-stage LDS-DMA implicit GEMM (gemm_glds.hip), bf16, N >= 128; variant codes 200 (128x128) / 201 (64x128) / 202 (256x128)
bool conv_glds_supported(const tfpp_conv_params& p, int dtype);
int conv_glds_variant(const tfpp_conv_params& p);
int conv_glds_bm(int variant);  // rows per M-tile
int conv_gemm_glds(const tfpp_conv_params& p, hipStream_t st);

// ping-pong LDS-DMA GEMM (gemm_pp.hip), bf16 pointwise layers with large M, N, K; variant codes 210 + configuration index
bool conv_pp_supported(const tfpp_conv_params& p, int dtype);
int conv_pp_variant(const tfpp_conv_params& p);
int conv_pp_splits(const tfpp_conv_params& p);
int conv_pp_bm(int variant);
int conv_gemm_pp(const tfpp_conv_params& p, hipStream_t st);

// 3x3 / stride 1 / pad 1 with the input tile staged once in LDS (conv3x3_halo.hip), bf16, n_g <= 64; variant codes 300 + FN
bool conv_halo_supported(const tfpp_conv_params& p, int dtype);
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
