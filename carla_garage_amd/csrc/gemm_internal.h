// Internal (non-ABI) declarations shared between the GEMM translation units.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tfpp.h"

// barrier-free direct-to-register implicit GEMM (gemm_direct.hip); variant code = 100 + FM*10 + FN
int conv_direct_variant(const tfpp_conv_params& p, int dtype);
int conv_gemm_direct(const tfpp_conv_params& p, int dtype, hipStream_t st);

// multi-stage LDS-DMA implicit GEMM (gemm_glds.hip), bf16, N >= 128; variant codes 200 (128x128) / 201 (64x128)
bool conv_glds_supported(const tfpp_conv_params& p, int dtype);
int conv_glds_variant(const tfpp_conv_params& p);
int conv_gemm_glds(const tfpp_conv_params& p, hipStream_t st);
