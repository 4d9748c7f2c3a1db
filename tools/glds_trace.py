#!/usr/bin/env python
"""Phase breakdown of one LDS-DMA GEMM launch (TFPP_GLDS_TRACE=1): setup / ring prologue / first tile landed / K loop / epilogue."""
import ctypes
import os
import sys

os.environ['TFPP_GLDS_TRACE'] = str(int(os.environ.get('TFPP_GLDS_TRACE', '1')) | 1)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402
from carla_garage_amd._lib import lib  # noqa: E402


def main():
  M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3072, 576, 576)))
  dt = torch.bfloat16
  x = (torch.rand(M, 1, 1, K, device='cuda') - 0.5).to(dt)
  w = (torch.rand(N, K, 1, 1, device='cuda') - 0.5) * 0.1
  wp = ops.pack_conv_weight(w, dt)
  y = torch.empty(M, 1, 1, N, device='cuda', dtype=dt)
  for _ in range(3):
    ops.conv_gemm(x, wp, y, B=M, Hs=1, Ws=1, Cs=K, Hd=1, Wd=1, Cd=N)
  torch.cuda.synchronize()
  nb = min(4096, ((M + 63) // 64) * ((N + 127) // 128))
  buf = np.zeros(nb * 6, dtype=np.uint64)
  slots = lib.raw('tfpp_debug_glds_trace')(buf.ctypes.data_as(ctypes.c_void_p), nb)
  t = buf.reshape(nb, 6).astype(np.int64)
  t = t[t[:, 5] > 0]
  t0 = t[:, 0].min()
  names = ['start', 'setup done', 'prologue issued', 'tile 0 landed', 'K loop done', 'end']
  print(f'M={M} K={K} N={N}: {len(t)} workgroups traced, slots={slots}; times in us relative to the first workgroup start')
  for k, nme in enumerate(names):
    col = (t[:, k] - t0) / 100.0
    print(f'  {nme:16s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f}')
  tm = np.median(t[:, 0]) + 0.5 * np.median(t[:, 5] - t[:, 0])
  for frac in (0.25, 0.5, 0.75):
    tq = t[:, 0].min() + frac * (t[:, 5].max() - t[:, 0].min())
    print(f'  workgroups resident at {int(frac * 100)} % of the launch: {int(((t[:, 0] <= tq) & (t[:, 5] > tq)).sum())}  (256 CUs)')
  nkt = (K + 31) // 32
  print(f'  K loop per k-step (median): {np.median(t[:, 4] - t[:, 3]) / 100.0 / nkt * 1000:.0f} ns for {nkt} k-steps; launch span {(t[:, 5].max() - t0) / 100.0:.1f} us')
  d = np.diff(t, axis=1) / 100.0
  print('  per-workgroup phase durations (median us):', ' '.join(f'{nme}={np.median(d[:, k]):.2f}' for k, nme in enumerate(['setup', 'issue', 'first-wait', 'kloop', 'epilogue'])))


if __name__ == '__main__':
  main()
