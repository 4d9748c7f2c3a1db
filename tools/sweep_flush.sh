#!/bin/bash
for fa in "144,256" "136,256" "160,256" "152,256" "144,256,304" "144,256,320" "80,144,256" "104,144,256" "144,216,256" "144,176,256" "148,256,304"; do
  ms=$(TFPP_SIDE_FLUSH_AT=$fa python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-roofline --no-inference --no-dropin 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "flush at $fa : $ms ms/step"
done
