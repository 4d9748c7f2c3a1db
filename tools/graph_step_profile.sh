#!/bin/bash
# rocprofv3 kernel trace of hipGraph-replayed training steps; prints span / busy time of the last step and writes the
# per-kernel table of that step to gpurun_out/graph_step_kernels.txt.  Run on the GPU box from the repo root.
# `bash tools/graph_step_profile.sh swin`: the same for BASELINE config 5 (Video-Swin LiDAR branch, bs = 4): files gpurun_out/swin_graph_step*.
set -e
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
PREFIX=graph_step
if [ "$1" = "swin" ]; then
  PREFIX=swin_graph_step
  TFPP_CONFIG5_NO_ROOFLINE=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $REPO/bench.py --config5-only --steps 4 > $REPO/gpurun_out/prof_bench.log 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-inference --no-roofline --no-dropin > $REPO/gpurun_out/prof_bench.log 2>&1
fi
t=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python - "$t" "$REPO/gpurun_out/${PREFIX}_kernels.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
step = rows[a:b]
t0 = min(int(r["Start_Timestamp"]) for r in step); t1 = max(int(r["End_Timestamp"]) for r in step)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
# union of busy intervals (streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
cov, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e: cov += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
cov += cur_e - cur_s
print("last step: %d kernels, span %.2f ms, sum of kernel durations %.2f ms, time with >=1 kernel running %.2f ms" % (len(step), (t1 - t0) / 1e6, busy / 1e6, cov / 1e6))
qkey = "Queue_Id" if "Queue_Id" in step[0] else None
if qkey:
    perq = collections.defaultdict(lambda: [0, 0, collections.defaultdict(float)])
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        q = perq[r[qkey]]; q[0] += 1; q[1] += d; q[2][r["Kernel_Name"][:60]] += d
    for qid, (n, busy_q, ks) in sorted(perq.items(), key=lambda kv: -kv[1][1]):
        top = ", ".join("%s %.2f" % (k[:40], v / 1e6) for k, v in sorted(ks.items(), key=lambda kv: -kv[1])[:6])
        print("queue %s: %d kernels, busy %.2f ms; top: %s" % (qid, n, busy_q / 1e6, top))
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    k = r["Kernel_Name"][:100]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2].replace("kernels.txt", "trace.csv"), "w") as out:  # the whole last step, one line per kernel, for offline timeline analysis
    out.write("queue,start_us,dur_us,kernel\n")
    for r in step:
        out.write("%s,%.2f,%.2f,%s\n" % (r[qkey] if qkey else "0", (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                      r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace(",", ";")[:70]))
import json
qb = {}
if qkey:
    for qid, (n, busy_q, ks) in perq.items():
        qb[str(qid)] = {"kernels": n, "busy_ms": round(busy_q / 1e6, 3)}
with open(sys.argv[2].replace("_kernels.txt", ".json"), "w") as out:
    # machine-readable summary of the same step: bench.py attaches it (graph-accurate launch durations) to its roofline objects
    json.dump({"what": "rocprofv3 --kernel-trace of one hipGraph-replayed training step (%s), tools/graph_step_profile.sh" % ("Video-Swin configuration, bs=4 bf16" if "swin" in sys.argv[2] else "bs=12 bf16"),
               "launches": len(step), "span_ms": round((t1 - t0) / 1e6, 3), "sum_kernel_ms": round(busy / 1e6, 3), "busy_union_ms": round(cov / 1e6, 3),
               "queues": qb, "kernels": {k: {"calls": v[0], "total_ms": round(v[1] / 1e6, 4), "avg_us": round(v[1] / v[0] / 1e3, 3)} for k, v in agg.items()}},
              out, indent=0)
with open(sys.argv[2], "w") as out:
    out.write("# kernel | calls | total_ms | avg_us   (one hipGraph-replayed training step, bs=12 bf16)\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write("%-102s %5d %9.3f ms %8.2f us\n" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
PY
