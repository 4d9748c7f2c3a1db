#!/bin/bash
# the weight-gradient lane on a CU subset (hipExtStreamCreateWithCUMask, n CUs of every XCD) -- eagerly and inside the captured step.
for mode in "--no-graph" ""; do
  for n in 0 8 16 24; do
    ms=$(TFPP_SIDE_CU_MASK=$n timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-inference --no-dropin $mode 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "side lane on $n of 32 CUs per XCD (0 = all) ${mode:-hipGraph replay}: $ms ms/step"
  done
done
