#!/bin/bash
# per-kernel durations of the GEMM families inside the hipGraph-replayed step (rocprofv3 kernel trace), ping-pong GEMM on / off
for v in 0 -1; do
  TFPP_GEMM_PP=$v bash tools/graph_step_profile.sh > gpurun_out/pp_graph_$v.txt 2>&1
  cp gpurun_out/graph_step_kernels.txt gpurun_out/pp_graph_kernels_$v.txt
  echo "== TFPP_GEMM_PP=$v"; head -3 gpurun_out/pp_graph_$v.txt; grep -E "conv_gemm_(pp|glds)_kernel" gpurun_out/pp_graph_kernels_$v.txt
done
