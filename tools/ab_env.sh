#!/bin/bash
# A/B of environment settings on one box: tools/ab_env.sh "VAR=a" "VAR=b" ...   (each run: bench.py training leg only, 30 steps; two rounds)
for round in 1 2; do
  for setting in "$@"; do
    ms=$(env $setting python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-inference --no-dropin 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$setting  $ms ms/step"
  done
done
