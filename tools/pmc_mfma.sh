#!/bin/bash
# MFMA-pipe utilisation from hardware counters: rocprofv3 --pmc passes (counters only, no tracing) over two eager training
# steps (bs = 12, bf16), aggregated per kernel family -> gpurun_out/pmc_mfma.json + .txt.
#   pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE   (MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * SIMDs))
#   pass B: SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32  (x 512 = FLOPs the matrix pipe executed)
# usage (GPU box, repo root): bash tools/pmc_mfma.sh
set -e
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcA /tmp/pmcB
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcA -- \
  python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline --no-dropin > $REPO/gpurun_out/pmc_mfma_A.log 2>&1 || echo "pass A failed"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pmcB -- \
  python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline --no-dropin > $REPO/gpurun_out/pmc_mfma_B.log 2>&1 || echo "pass B failed"
rm -rf /tmp/pmcC
timeout 400 rocprofv3 --pmc MfmaUtil --kernel-include-regex "conv_gemm|conv_wgrad|conv3x3_halo|wgrad3x3|attn_|bgemm" --output-format csv -d /tmp/pmcC -- \
  python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline --no-dropin > $REPO/gpurun_out/pmc_mfma_C.log 2>&1 || echo "pass C failed"
python - "$REPO" <<'PY'
import csv, glob, json, sys, collections, re
repo = sys.argv[1]
SIMDS = 256 * 4
def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); seen = set()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).replace('void ', '')[:90]
            out[k][r['Counter_Name']] += float(r['Counter_Value'])
            key = (k, r.get('Dispatch_Id'))
            if key not in seen:
                seen.add(key); n[k] += 1
    return out, n
A, nA = load('/tmp/pmcA'); B, nB = load('/tmp/pmcB'); Cc, nC = load('/tmp/pmcC')
rows = {}
for k in A:
    a = A[k]; busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0); gui = a.get('GRBM_GUI_ACTIVE', 0.0); sq = a.get('SQ_BUSY_CYCLES', 0.0)
    if busy <= 0: continue
    b = B.get(k, {})
    util = Cc.get(k, {}).get('MfmaUtil')
    rows[k] = {'dispatches': nA[k], 'mfma_busy_cycles': busy, 'grbm_gui_active': gui, 'sq_busy_cycles': sq,
               # rocprofv3's derived metric (reduce(SQ_VALU_MFMA_BUSY_CYCLES,sum) / (reduce(GRBM_GUI_ACTIVE,max) * SIMD_NUM) * 100), mean over the dispatches
               'MfmaUtil_pct': round(util / nC[k], 2) if util is not None and nC.get(k) else None,
               # the same ratio from the raw counters of pass A, where the CSV sums GRBM_GUI_ACTIVE over the 8 XCDs (hence x 8 against MfmaUtil)
               'mfma_util_pct_of_all_simds': round(100.0 * busy / (gui * SIMDS), 2) if gui else None,
               'mfma_flops_bf16': b.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512, 'mfma_flops_f32': b.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) * 512}
tot_busy = sum(r['mfma_busy_cycles'] for r in rows.values()); tot_gui = sum(r['grbm_gui_active'] for r in rows.values())
wnum = sum(r['MfmaUtil_pct'] * r['grbm_gui_active'] for r in rows.values() if r['MfmaUtil_pct'] is not None)
wden = sum(r['grbm_gui_active'] for r in rows.values() if r['MfmaUtil_pct'] is not None)
res = {'what': 'rocprofv3 --pmc, two eager training steps bs=12 bf16 (counters serialise the kernels: per-kernel values are for a kernel ALONE on the chip)',
       'formula': 'MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) (rocprofv3 derived metric MfmaUtil); flops = SQ_INSTS_VALU_MFMA_MOPS_* x 512',
       'all_mfma_kernels': {'MfmaUtil_pct_time_weighted': round(wnum / wden, 2) if wden else None,
                            'mfma_flops_bf16': sum(r['mfma_flops_bf16'] for r in rows.values()), 'mfma_flops_f32': sum(r['mfma_flops_f32'] for r in rows.values())},
       'kernels': dict(sorted(rows.items(), key=lambda kv: -kv[1]['mfma_busy_cycles']))}
json.dump(res, open(repo + '/gpurun_out/pmc_mfma.json', 'w'), indent=1)
with open(repo + '/gpurun_out/pmc_mfma.txt', 'w') as f:
    f.write('# kernel | dispatches | MfmaUtil %% (rocprofv3 derived metric, mean over dispatches) | raw busy/(gui_sum x 1024) %% | bf16 MFMA GFLOP | fp32 MFMA GFLOP\n')
    for k, r in res['kernels'].items():
        f.write('%-92s %5d %7s %7s %10.2f %10.2f\n' % (k, r['dispatches'], r['MfmaUtil_pct'], r['mfma_util_pct_of_all_simds'], r['mfma_flops_bf16'] / 1e9, r['mfma_flops_f32'] / 1e9))
    f.write('# all kernels with MFMA work, weighted by GPU-active cycles: MfmaUtil %s %%\n' % res['all_mfma_kernels']['MfmaUtil_pct_time_weighted'])
print(open(repo + '/gpurun_out/pmc_mfma.txt').read()[:3000])
PY
