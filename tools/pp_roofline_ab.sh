#!/bin/bash
# roofline of the fusion-transformer linears BY LAYER (bench.py, KernelProfiler group) + step time: ring kernels / ping-pong GEMM plans
for setting in "$@"; do
  env $setting python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-inference --no-dropin 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline_fusion_linears', {})
print('$setting', d['ms_per_step'], 'ms/step; fusion linears', r.get('achieved'), 'TFLOP/s frac', r.get('frac'), 'avg us', r.get('avg_launch_us'), 'launches', r.get('launches_per_step'), r.get('kernels'))"
done
