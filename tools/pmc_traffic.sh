#!/bin/bash
# HBM-side traffic of one kernel family from rocprofv3 PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE, counters only,
# no tracing), restricted to the kernels matching $1 over two eager training steps.  Writes gpurun_out/pmc_<counter>.csv summaries.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh conv_wgrad_glds_kernel
set -e
REPO=$(pwd)
KREGEX=${1:-conv_wgrad_glds_kernel}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 250 rocprofv3 --pmc $C --kernel-include-regex "$KREGEX" --output-format csv -d /tmp/pmc_$C -- \
    python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline > $REPO/gpurun_out/pmc_$C.log 2>&1 || echo "pass $C failed/timeout"
  f=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -1)
  python - "$f" "$C" "$KREGEX" <<'PY'
import csv, sys, collections
f, cname, kre = sys.argv[1:4]
if not f:
    print(cname, "no output"); sys.exit(0)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != cname: continue
    k = r["Kernel_Name"][:80]
    agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, (n, v) in agg.items():
    print("%s %s: launches %d, mean raw counter %.1f per launch" % (cname, k, n, v / n))
PY
done
