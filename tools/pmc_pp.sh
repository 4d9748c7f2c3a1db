#!/bin/bash
# PMC counters of the ping-pong GEMM (or the ring kernel it replaces) on one micro-benchmark shape: separate counter-only rocprofv3 passes.
# usage (GPU box, repo root): bash tools/pmc_pp.sh <cfg of tfpp_gemm_pp_config> <shape pattern> [kernel regex]  -> gpurun_out/pmc_pp_<cfg>_<shape>.txt
set -e
REPO=$(pwd)
CFG=$1; ONLY=$2; KRE=${3:-conv_gemm_(pp|glds)_kernel}
OUT=$REPO/gpurun_out/pmc_pp_${CFG}_${ONLY}.txt
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
pass() {
  local name=$1; shift
  rm -rf /tmp/pmcp_$name
  timeout 200 rocprofv3 --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d /tmp/pmcp_$name -- \
    python $REPO/tools/gemm_pp_micro.py --only "$ONLY" --eager --iters 3 --cfgs=$CFG > /tmp/pmcp_$name.log 2>&1 || echo "pass $name failed" >> $OUT
  python - /tmp/pmcp_$name >> $OUT <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:60], r['Counter_Name'])
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(agg.items()):
    print('%-62s %-28s launches %3d  mean %.5g' % (k, c, n, v / n))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass misc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU
cat $OUT
