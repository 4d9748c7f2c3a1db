#!/bin/bash
# HBM-side traffic of EVERY kernel of one eager training step (bs = 12 bf16), per kernel name: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# two counter-only passes (no tracing), summed per kernel name.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B,
# MI355X_MICROARCH.md).  Output: gpurun_out/pmc_all.txt  (kernel | launches | fetch MB | write MB | MB per launch), sorted by bytes.
# usage (GPU box, repo root): bash tools/pmc_all.sh [extra bench.py flags]
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmca_$C
  timeout 400 rocprofv3 --pmc $C --output-format csv -d /tmp/pmca_$C -- \
    python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline --no-dropin "$@" > $REPO/gpurun_out/pmca_$C.log 2>&1 || echo "pass $C failed/timeout"
done
python - "$REPO" <<'PY'
import csv, glob, sys, collections
repo = sys.argv[1]
tab = collections.defaultdict(lambda: [0, 0.0, 0.0])
for cname, col, mul in (('FETCH_SIZE', 1, 2048.0), ('WRITE_SIZE', 2, 1024.0)):
    fs = glob.glob(f'/tmp/pmca_{cname}/**/*counter_collection.csv', recursive=True)
    if not fs:
        print('no output for', cname); continue
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name') != cname:
            continue
        k = r['Kernel_Name']
        if col == 1:
            tab[k][0] += 1
        tab[k][col] += float(r['Counter_Value']) * mul
rows = sorted(tab.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))
tot_f = sum(v[1] for _, v in rows); tot_w = sum(v[2] for _, v in rows)
with open(repo + '/gpurun_out/pmc_all.txt', 'w') as f:
    f.write('# launches counted over warm-up + timed eager step (2 steps + set-up); MB = 1e6 bytes\n')
    f.write('# total fetch %.1f MB, write %.1f MB\n' % (tot_f / 1e6, tot_w / 1e6))
    for k, (n, fb, wb) in rows:
        f.write('%-110s %5d  fetch %9.1f MB  write %9.1f MB  per launch %8.2f / %8.2f MB\n' % (k[:110], n, fb / 1e6, wb / 1e6, fb / 1e6 / max(n, 1), wb / 1e6 / max(n, 1)))
import json, time
json.dump({'_comment': 'HBM-side bytes of EVERY kernel over two eager training steps (bs = 12 bf16): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate counter-only passes, FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md); written by tools/pmc_all.sh. bench.py falls back to this table for kernel families without an entry in pmc_traffic.json.',
           'measured': time.strftime('%Y-%m-%d'), 'kernels': {k[:110].strip(): {'launches': n, 'fetch_bytes': fb, 'write_bytes': wb} for k, (n, fb, wb) in rows}},
          open(repo + '/gpurun_out/pmc_all_kernels.json', 'w'), indent=1)
print(open(repo + '/gpurun_out/pmc_all.txt').read()[:6000])
PY
