#!/bin/bash
# A/B of environment settings on one box, training leg only: tools/ab_quick.sh "VAR=a" "VAR=b,OTHER=c" ...  (two rounds, 30 timed steps each)
for round in 1 2; do
  for setting in "$@"; do
    env $(echo "$setting" | tr ',' ' ') timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin --no-inference 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$setting  %.3f ms/step' % d['ms_per_step'])"
  done
done
