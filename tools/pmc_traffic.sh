#!/bin/bash
# HBM-side traffic of one kernel family from rocprofv3 PMC counters (separate passes for FETCH_SIZE and WRITE_SIZE, counters only,
# no tracing -- gpurun refuses --pmc combined with trace domains), restricted to the kernels matching $1 over two eager training
# steps.  Merges the result into profiles/pmc_traffic.json under the bench.py family name $2 together with the sha of the kernel
# source file $3, so bench.py can refuse the entry once that file changes.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh conv_wgrad_glds_kernel 'conv_wgrad<bf16,glds128x128>' gemm_wgrad_glds.hip
set -e
REPO=$(pwd)
KREGEX=${1:-conv_wgrad_glds_kernel}
FAMILY=${2:-conv_wgrad<bf16,glds64x64>}
SRCFILE=${3:-gemm_wgrad_glds.hip}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 250 rocprofv3 --pmc $C --kernel-include-regex "$KREGEX" --output-format csv -d /tmp/pmc_$C -- \
    python $REPO/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-inference --no-roofline --no-dropin > $REPO/gpurun_out/pmc_$C.log 2>&1 || echo "pass $C failed/timeout"
done
python - "$REPO" "$KREGEX" "$FAMILY" "$SRCFILE" <<'PY'
import csv, glob, hashlib, json, os, sys, time
repo, kre, family, srcfile = sys.argv[1:5]
res = {}
for cname in ('FETCH_SIZE', 'WRITE_SIZE'):
    fs = glob.glob(f'/tmp/pmc_{cname}/**/*counter_collection.csv', recursive=True)
    n, tot = 0, 0.0
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if r.get('Counter_Name') == cname:
                n += 1; tot += float(r['Counter_Value'])
    res[cname] = (n, tot / n if n else None)
    print(cname, kre, 'launches', n, 'mean raw counter per launch', res[cname][1])
path = os.path.join(repo, 'profiles', 'pmc_traffic.json')
try:
    db = json.load(open(path))
except (OSError, ValueError):
    db = {}
sha = hashlib.sha256(open(os.path.join(repo, 'carla_garage_amd', 'csrc', srcfile), 'rb').read()).hexdigest()[:16]
nf, f = res['FETCH_SIZE']; nw, w = res['WRITE_SIZE']
if f is not None:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 (MI355X_MICROARCH.md, HBM)
    db[family] = {'kernel_regex': kre, 'launches': nf, 'fetch_raw_kb_per_launch': f, 'write_raw_kb_per_launch': w,
                  'fetch_bytes_per_launch': int(f * 1024 * 2), 'write_bytes_per_launch': int(w * 1024) if w is not None else None,
                  'source_file': srcfile, 'source_sha16': sha, 'measured': time.strftime('%Y-%m-%d') + ', 2 eager train steps bs=12 bf16'}
    json.dump(db, open(path, 'w'), indent=1)
    print('updated', path, family)
PY
