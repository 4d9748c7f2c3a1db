#!/bin/bash
# PMC counters of ONE kernel on one micro-benchmark shape (tools/gemm_micro.py): separate counter-only rocprofv3 passes
# (SQ: 8 slots, TCC: 4, FETCH_SIZE alone).  usage (GPU box, repo root):
#   bash tools/pmc_micro.sh <kernel regex> <gemm_micro --only pattern> [--wgrad]     -> gpurun_out/pmc_micro_<tag>.txt
set -e
REPO=$(pwd)
KRE=$1; ONLY=$2; EXTRA=$3
TAG=$(echo "${KRE}_${ONLY}${EXTRA}" | tr -c 'A-Za-z0-9_\n' '_')
OUT=$REPO/gpurun_out/pmc_micro_$TAG.txt
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $OUT
pass() {
  local name=$1; shift
  rm -rf /tmp/pmcm_$name
  timeout 200 rocprofv3 --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d /tmp/pmcm_$name -- \
    python $REPO/tools/gemm_micro.py --only "$ONLY" $EXTRA --eager --iters 3 > /tmp/pmcm_$name.log 2>&1 || echo "pass $name failed" >> $OUT
  python - /tmp/pmcm_$name >> $OUT <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:70], r['Counter_Name'])
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(agg.items()):
    print('%-72s %-28s launches %3d  mean %.4g' % (k, c, n, v / n))
PY
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass waves SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA
cat $OUT
