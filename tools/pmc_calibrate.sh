#!/bin/bash
# rocprofv3 FETCH_SIZE / WRITE_SIZE of the known-byte launches of tools/pmc_calibrate.py (counter-only passes); prints raw KB per launch
# next to the known byte counts and writes gpurun_out/pmc_calibration.json.  Run on the GPU box from the repo root.
set -e
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/cal_$C -- python $REPO/tools/pmc_calibrate.py > $REPO/gpurun_out/pmc_cal_$C.log 2>&1 || echo "pass $C failed"
done
python - "$REPO" <<'PY'
import csv, glob, json, sys, collections
repo = sys.argv[1]
import os
M, K, N = (int(os.environ.get(k, d)) for k, d in (('CAL_M', 3840), ('CAL_K', 1512), ('CAL_N', 6048)))
known = {'fillBuffer': (0, 256 * 1024 * 1024 * 4), 'affine_act': ((1 << 20) * 256 * 2, (1 << 20) * 256 * 2), 'conv_gemm_glds': ((M * K + K * N) * 2, M * N * 2)}
out = {}
for ci, cname in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
    fs = glob.glob(f'/tmp/cal_{cname}/**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name') == cname:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, vals in agg.items():
        for key, kb in known.items():
            if key in k and 'pack' not in k:
                v = sum(vals[-3:]) / len(vals[-3:])
                e = out.setdefault(key, {})
                e[cname + '_raw_kb'] = v
                e[cname + '_known_bytes'] = kb[ci]
                e[cname + '_bytes_per_raw_kb'] = (kb[ci] / v) if v else None
                print(cname, key, 'raw KB/launch %.1f' % v, 'known bytes', kb[ci], 'bytes per raw KB %.1f' % ((kb[ci] / v) if v else 0))
json.dump(out, open(repo + '/gpurun_out/pmc_calibration.json', 'w'), indent=1)
PY
