#!/usr/bin/env python
"""Round-3 driver of tools/replay_bisect.py on the GPU box: confirms the replay non-reproducibility at TFPP_SIDE_BATCH=128 without any
instrumentation, then narrows it down with the per-event hash tables (full table first, then only a window around the first differing
event so that the extra hash launches perturb the schedule as little as possible).  Everything is written under gpurun_out/r03/."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'r03')
os.makedirs(OUT, exist_ok=True)


def run(name, cmd, env):
  e = dict(os.environ)
  e.update(env)
  path = os.path.join(OUT, name + '.txt')
  with open(path, 'w') as f:
    f.write(f'# {env} {" ".join(cmd)}\n')
    f.flush()
    try:
      subprocess.run(cmd, cwd=ROOT, env=e, stdout=f, stderr=subprocess.STDOUT, timeout=420, check=False)
    except subprocess.TimeoutExpired:
      f.write('\nTIMEOUT\n')
  txt = open(path).read()
  print(f'===== {name}\n' + '\n'.join(txt.splitlines()[-45:]), flush=True)
  return txt


def parse(txt):
  m = re.search(r'events (\d+) .*hash tables deviating (\d+), gradient arenas deviating (\d+)', txt)
  if not m:
    return None
  first = re.search(r'first differing event per deviating replay: \[([^\]]*)\]', txt)
  firsts = [int(v) for v in first.group(1).split(',')] if first and first.group(1).strip() else []
  fwd = re.search(r'forward events (\d+), backward events (\d+)', txt)
  return dict(events=int(m.group(1)), dev=int(m.group(2)), gdev=int(m.group(3)), firsts=firsts, fwd=int(fwd.group(1)) if fwd else None)


def main():
  py = sys.executable
  sb = os.environ.get('DIAG_SIDE_BATCH', '128')
  run('diff_plain', [py, 'tools/replay_diff.py'], {'TFPP_SIDE_BATCH': sb, 'DIAG_TOP': '4'})
  full = parse(run('bisect_full', [py, 'tools/replay_bisect.py', '30'], {'TFPP_SIDE_BATCH': sb, 'TFPP_DEBUG_NODE_HASH': '1'}))
  run('bisect_control32', [py, 'tools/replay_bisect.py', '20'], {'TFPP_SIDE_BATCH': '32', 'TFPP_DEBUG_NODE_HASH': '1'})
  if full is None:
    return
  n = full['events']
  if full['dev'] and full['firsts']:
    f0 = min(full['firsts'])
    for k, (lo, hi) in enumerate(((max(0, f0 - 40), f0 + 40), (max(0, f0 - 6), f0 + 6))):
      run(f'bisect_window{k}', [py, 'tools/replay_bisect.py', '30'], {'TFPP_SIDE_BATCH': sb, 'TFPP_DEBUG_NODE_HASH': '1', 'TFPP_DEBUG_NODE_HASH_RANGE': f'{lo}:{hi}'})
  else:  # the hash launches moved the overlap: hash only parts of the step
    half = n // 2
    for k, (lo, hi) in enumerate(((half, n), (half, half + (n - half) // 3), (half + (n - half) // 3, half + 2 * (n - half) // 3), (0, half))):
      r = parse(run(f'bisect_part{k}', [py, 'tools/replay_bisect.py', '30'], {'TFPP_SIDE_BATCH': sb, 'TFPP_DEBUG_NODE_HASH': '1', 'TFPP_DEBUG_NODE_HASH_RANGE': f'{lo}:{hi}'}))
      if r and r['dev'] and r['firsts']:
        f0 = min(r['firsts'])
        run(f'bisect_part{k}_window', [py, 'tools/replay_bisect.py', '30'],
            {'TFPP_SIDE_BATCH': sb, 'TFPP_DEBUG_NODE_HASH': '1', 'TFPP_DEBUG_NODE_HASH_RANGE': f'{max(0, f0 - 6)}:{f0 + 6}'})
        break


if __name__ == '__main__':
  main()
