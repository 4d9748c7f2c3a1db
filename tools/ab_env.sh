#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh "VAR=a" "VAR=b" ...   (each run: bench.py training leg + the bs=1 forward; two rounds).
# Separate several variables of one setting with commas: "A=1,B=2".
for round in 1 2; do
  for setting in "$@"; do
    env $(echo "$setting" | tr ',' ' ') python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); f=d.get('fwd_ms_per_frame') or {}; print('$setting  %.3f ms/step  fwd bs=1 bf16 %s fp32 %s' % (d['ms_per_step'], f.get('bf16_hipgraph'), f.get('fp32_hipgraph')))"
  done
done
