#!/usr/bin/env python
"""Compare two per-kernel tables written by tools/graph_step_profile.sh (e.g. the captured step with and without the weight-gradient lane:
TFPP_DEBUG_SKIP_SIDE_WORK=1): which kernels of the dY chain get slower when the lane runs beside them?
  python tools/kernel_table_diff.py a_kernels.txt b_kernels.txt"""
import re
import sys


def load(path):
  out = {}
  for line in open(path, encoding='utf-8'):
    m = re.match(r'(.+?)\s+(\d+)\s+([\d.]+) ms\s+([\d.]+) us\s*$', line)
    if m:
      out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
  return out


def main():
  a, b = load(sys.argv[1]), load(sys.argv[2])
  rows = []
  for k in a:
    if k in b and a[k][0] == b[k][0]:
      rows.append((b[k][1] - a[k][1], k, a[k], b[k]))
  rows.sort(reverse=True)
  print(f'# kernels with equal call counts in both tables; total_ms a -> b, avg_us a -> b  (a = {sys.argv[1]}, b = {sys.argv[2]})')
  tot = 0.0
  for d, k, x, y in rows[:45]:
    print(f'{d:+7.3f} ms  {x[0]:4d} calls  {x[2]:7.2f} -> {y[2]:7.2f} us  {k[:110]}')
  for d, *_ in rows:
    tot += d
  print(f'sum over {len(rows)} common kernels: {tot:+.3f} ms;  only in a: {sum(v[1] for k, v in a.items() if k not in b or a[k][0] != b[k][0]):.3f} ms, only in b: {sum(v[1] for k, v in b.items() if k not in a or a[k][0] != b[k][0]):.3f} ms')


if __name__ == '__main__':
  main()
