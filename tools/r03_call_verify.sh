#!/bin/bash
# round 3: the fix of the co-runner-dependent result (no packed FP32 instructions) under the conditions that used to fail
mkdir -p gpurun_out/r03
for sb in 128 96; do
  TFPP_SIDE_BATCH=$sb TFPP_DEBUG_NODE_HASH=1 timeout 300 python tools/replay_bisect.py 40 > gpurun_out/r03/fixed_bisect_$sb.txt 2>&1
  grep "events" gpurun_out/r03/fixed_bisect_$sb.txt
done
TFPP_SIDE_BATCH=128 timeout 600 python tools/stress_step.py --replays 400 --eager 60 > gpurun_out/r03/fixed_stress_128.txt 2>&1; tail -2 gpurun_out/r03/fixed_stress_128.txt
TFPP_SIDE_BATCH=96 timeout 600 python tools/stress_step.py --replays 200 --eager 0 > gpurun_out/r03/fixed_stress_96.txt 2>&1; tail -2 gpurun_out/r03/fixed_stress_96.txt
for sb in 32 128 32 128; do
  TFPP_SIDE_BATCH=$sb timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-inference 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side batch $sb', d['ms_per_step'], d['value'])"
done
