#!/bin/bash
# Round-6 evidence in one gpurun call (GPU box, repo root): replayed-step traces (default configuration and Video-Swin), lane timeline,
# PMC traffic of the roofline kernel families, per-kernel traffic of the whole step, MFMA counters.  Copies of what is to be judged go to
# profiles/ afterwards (by hand).
mkdir -p gpurun_out
bash tools/graph_step_profile.sh > gpurun_out/graph_step_summary.txt 2>&1
bash tools/graph_step_profile.sh swin > gpurun_out/swin_graph_step_summary.txt 2>&1
python tools/lane_timeline.py --dump gpurun_out/lane_stamps.txt > gpurun_out/lane_timeline.txt 2>&1
bash tools/pmc_traffic.sh 'conv_wgrad_glds_group_kernel<256' 'conv_wgrad<bf16,group256>' gemm_wgrad_glds.hip > gpurun_out/pmc_traffic_group256.log 2>&1
bash tools/pmc_traffic.sh 'conv_wgrad_glds_group_kernel<128' 'conv_wgrad<bf16,group128>' gemm_wgrad_glds.hip > gpurun_out/pmc_traffic_group128.log 2>&1
bash tools/pmc_traffic.sh 'conv_wgrad_glds_group_kernel<64' 'conv_wgrad<bf16,group64>' gemm_wgrad_glds.hip > gpurun_out/pmc_traffic_group64.log 2>&1
bash tools/pmc_traffic.sh 'conv_gemm_glds_kernel<256' 'conv_gemm<bf16,glds256x128>' gemm_glds.hip > gpurun_out/pmc_traffic_glds256.log 2>&1
bash tools/pmc_traffic.sh 'conv_gemm_glds_kernel<128, 128' 'conv_gemm<bf16,glds128x128>' gemm_glds.hip > gpurun_out/pmc_traffic_glds128.log 2>&1
bash tools/pmc_traffic.sh 'conv_gemm_glds_kernel<64, 128' 'conv_gemm<bf16,glds64x128>' gemm_glds.hip > gpurun_out/pmc_traffic_glds64.log 2>&1
bash tools/pmc_traffic.sh 'wgrad3x3_halo_kernel<2, 3, true' 'wgrad3x3<bf16,halo2x3bn>' wgrad3x3_halo.hip > gpurun_out/pmc_traffic_wgrad3x3.log 2>&1
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/pmc_all.sh > gpurun_out/pmc_all.log 2>&1
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.log 2>&1
