#!/usr/bin/env python
"""Register / scratch / occupancy table of every gfx950 kernel in carla_garage_amd/csrc (hipcc -S, no GPU needed).
  python tools/kernel_resources.py [--scratch-only] [file.hip ...]
A non-zero ScratchSize in a hot kernel means private-memory traffic (it shows up in FETCH_SIZE / WRITE_SIZE as bytes the algorithm never
asked for: round 2 found 128 B/thread of it in the GEMM epilogues); TotalNumVgprs > 128 in a 512-thread kernel means one workgroup per CU."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
  try:
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
    return dict(zip(names, out))
  except Exception:  # pylint: disable=broad-except
    return {n: n for n in names}


_ASM = {}


def assembly(path):
  """gfx950 assembly of one source, compiled with the product's flags (cached per process)."""
  if path not in _ASM:
    with tempfile.TemporaryDirectory() as td:
      s_path = os.path.join(td, 'k.s')
      sys.path.insert(0, ROOT)
      from carla_garage_amd._lib import HIPCC_FLAGS  # the flags the product is built with
      r = subprocess.run(['hipcc'] + HIPCC_FLAGS + ['-S', '--cuda-device-only', '-DTFPP_SOURCE_HASH=0', path, '-o', s_path], capture_output=True, text=True)
      if r.returncode:
        raise RuntimeError(r.stderr[-2000:])
      _ASM[path] = open(s_path).read()
  return _ASM[path]


def table(path):
  s = assembly(path)
  rows = []
  # per kernel: .amdhsa_kernel NAME ... ; NumVgprs / NumAgprs / TotalNumVgprs / ScratchSize ... ; LDSByteSize ... ; Occupancy (in this order)
  for blk in s.split('.amdhsa_kernel ')[1:]:
    name = blk.split(None, 1)[0]
    g = lambda key: int(re.search(r'; ' + key + r': (\d+)', blk).group(1))
    rows.append((name, g('NumVgprs'), g('NumAgprs'), g('TotalNumVgprs'), g('ScratchSize'), g('Occupancy'), g('LDSByteSize')))
  return rows


def packed_fp32_instructions(path):
  """{kernel: count} of packed FP32 VALU instructions (v_pk_*_f32) in the gfx950 code of ``path`` -- there must be none (carla_garage_amd/_lib.py)."""
  out, cur = {}, None
  for line in assembly(path).split('\n'):
    m = re.match(r'^(_Z\w+):', line)
    if m:
      cur = m.group(1)
    elif cur and re.search(r'\bv_pk_\w+_f32\b', line):
      out[cur] = out.get(cur, 0) + 1
  return out


def main():
  args = [a for a in sys.argv[1:] if not a.startswith('--')]
  scratch_only = '--scratch-only' in sys.argv
  files = args or sorted(glob.glob(os.path.join(ROOT, 'carla_garage_amd', 'csrc', '*.hip')))
  bad = 0
  for f in files:
    rows = table(f)
    names = demangle([r[0] for r in rows])
    for name, v, a, t, sc, occ, lds in rows:
      if scratch_only and sc == 0:
        continue
      bad += sc > 0
      print('%-24s %-100s vgpr %3d agpr %3d total %3d scratch %4d occupancy %d static-lds %d' % (os.path.basename(f), names[name][:100], v, a, t, sc, occ, lds))
  return 1 if (scratch_only and bad) else 0


if __name__ == '__main__':
  sys.exit(main())
