#!/bin/bash
# Round evidence in one go (GPU box, repo root): default bench line, rocprofv3 --kernel-trace --stats of the same command, the
# hipGraph step kernel table, PMC traffic of the dominant kernel families, MFMA-pipe counters, isolated GEMM rates, the stress run, the
# bs = 1 inference launch list, the PCIe-inclusive step.  Everything lands in gpurun_out/ with the prefix $1 (default r05); copy what should
# be judged into profiles/.
P=${1:-r05}
R=$(pwd)
mkdir -p gpurun_out
python bench.py > gpurun_out/${P}_bench_default.json 2> gpurun_out/${P}_bench_default.err
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $R/bench.py --no-cpu-baseline --no-inference --no-dropin > $R/gpurun_out/${P}_bench_under_rocprof.json 2> /dev/null; f=$(find /tmp/st -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${P}_rocprofv3_kernel_stats_bench_default.csv)
bash tools/graph_step_profile.sh > gpurun_out/${P}_graph_step_summary.txt 2>&1
cp gpurun_out/graph_step_kernels.txt gpurun_out/${P}_graph_step_kernels_bs12_bf16.txt
cp gpurun_out/graph_step.json gpurun_out/${P}_graph_step.json
bash tools/pmc_traffic.sh "conv_gemm_kernel<unsigned short, 128, 32" "conv_gemm<bf16,128x32>" gemm_kernels.hip > gpurun_out/${P}_pmc_128x32.txt 2>&1
bash tools/pmc_traffic.sh "conv_wgrad_glds_group_kernel<128, ?128" "conv_wgrad_glds_group128" gemm_wgrad_glds.hip > gpurun_out/${P}_pmc_wgrad_group128.txt 2>&1
bash tools/pmc_traffic.sh "conv_gemm_glds_kernel<256, ?128" "conv_gemm<bf16,glds256x128>" gemm_glds.hip > gpurun_out/${P}_pmc_glds256.txt 2>&1
bash tools/pmc_traffic.sh "conv_gemm_glds_kernel<128, ?128" "conv_gemm<bf16,glds128x128>" gemm_glds.hip > gpurun_out/${P}_pmc_glds128.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/${P}_pmc_traffic.json
bash tools/pmc_mfma.sh > gpurun_out/${P}_pmc_mfma.log 2>&1
cp gpurun_out/pmc_mfma.json gpurun_out/${P}_pmc_mfma.json; cp gpurun_out/pmc_mfma.txt gpurun_out/${P}_pmc_mfma.txt
python bench.py --kernel-table --no-cpu-baseline --no-inference --no-dropin > gpurun_out/${P}_bench_table.json 2> gpurun_out/${P}_kernel_table_bs12_bf16.txt
{ echo "# forward"; python tools/gemm_micro.py --iters 10; echo "# data gradient"; python tools/gemm_micro.py --iters 10 --dgrad; echo "# weight gradient"; python tools/gemm_micro.py --iters 10 --wgrad; } 2>&1 | grep -v amdgpu > gpurun_out/${P}_gemm_micro.txt
timeout 600 python tools/stress_step.py > gpurun_out/${P}_stress_step.log 2>&1
bash tools/infer_trace.sh bf16 > /dev/null 2>&1; cp gpurun_out/infer_kernels_bf16.txt gpurun_out/${P}_infer_kernels_bs1_bf16.txt
python tools/h2d_step_time.py 2> /dev/null | tail -1 > gpurun_out/${P}_h2d_step_time.json
tail -c 600 gpurun_out/${P}_bench_default.json; echo; cat gpurun_out/${P}_graph_step_summary.txt | head -3; tail -4 gpurun_out/${P}_pmc_mfma.txt; tail -2 gpurun_out/${P}_stress_step.log | cut -c1-400
python tools/lane_timeline.py > gpurun_out/${P}_lane_timeline.txt 2>&1
python tools/chain_cost.py > gpurun_out/${P}_chain_cost.txt 2>&1
