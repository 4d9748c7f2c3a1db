#!/bin/bash
# fork points of the weight-gradient lane as fractions of the backward pass: tools/sweep_forks.sh "0.43,0.77,0.95" "0.45,0.77,0.95" ...
for round in 1 2; do
  for f in "$@"; do
    ms=$(TFPP_SIDE_FORKS=$f python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-inference --no-dropin 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "forks $f : $ms ms/step"
  done
done
