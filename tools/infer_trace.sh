#!/bin/bash
# Launch list of ONE eager bs=1 eval forward (last of N) from a rocprofv3 kernel trace -> gpurun_out/infer_kernels_$1.txt
set -e
REPO=$(pwd); DT=${1:-bf16}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/inf
TFPP_EVAL_GRAPH_AFTER=-1 rocprofv3 --kernel-trace --output-format csv -d /tmp/inf -- python $REPO/tools/infer_trace.py $DT 4 > $REPO/gpurun_out/infer_trace_$DT.log 2>&1
t=$(find /tmp/inf -name "*kernel_trace.csv" | head -1)
python - "$t" "$REPO/gpurun_out/infer_kernels_$DT.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# forwards are separated by host synchronisations: split at the 3 largest idle gaps of the tail
ends = [int(r["End_Timestamp"]) for r in rows]; starts = [int(r["Start_Timestamp"]) for r in rows]
gaps = sorted(((starts[i + 1] - ends[i], i) for i in range(len(rows) - 1)), reverse=True)[:3]
cut = max(i for _, i in gaps) + 1
step = rows[cut:]
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = r["Kernel_Name"][:110]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2], "w") as out:
    out.write("# one eager eval forward bs=1: %d launches, sum of kernel time %.3f ms\n" % (len(step), sum(v[1] for v in agg.values()) / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        out.write("%-112s %5d %9.3f ms %8.2f us\n" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
    out.write("# launch order\n")
    for r in step:
        out.write("%s\n" % r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:90])
print(open(sys.argv[2]).readline())
PY
