#!/usr/bin/env python
"""Known-byte launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reports half of a
wide coalesced read, WRITE_SIZE is uncalibrated).  Each kernel below moves a byte count that is known exactly and larger than the 256 MiB
Infinity Cache where that matters; tools/pmc_calibrate.sh runs this under `rocprofv3 --pmc` (one counter per pass) and prints raw counter /
known bytes per kernel, which is the correction tools/pmc_traffic.sh and bench.py apply.
  zero        write-only, 16 B per lane           : 1 GiB written
  affine_act  read + write, 16 B per lane (bf16)  : 512 MiB read, 512 MiB written
  conv_gemm   3840 x 6048 x 1512 bf16 (glds128x128): C = 46.45 MB written exactly once through the kernel's own epilogue store pattern;
              A + B = 29.9 MB is the minimum read"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402


def main():
  dev = 'cuda'
  z = torch.empty(256 * 1024 * 1024, device=dev, dtype=torch.float32)
  x = torch.randn(1 << 20, 256, device=dev).to(torch.bfloat16)
  y = torch.empty_like(x)
  sc = torch.ones(256, device=dev)
  sh = torch.zeros(256, device=dev)
  M, K, N = (int(os.environ.get(k, d)) for k, d in (('CAL_M', 3840), ('CAL_K', 1512), ('CAL_N', 6048)))  # default: the fusion MLP of stage 4
  a = (torch.rand(M, 1, 1, K, device=dev) - 0.5).to(torch.bfloat16)
  w = ops.pack_conv_weight((torch.rand(N, K, 1, 1, device=dev) - 0.5) * 0.1, torch.bfloat16)
  c = torch.empty(M, 1, 1, N, device=dev, dtype=torch.bfloat16)
  for _ in range(5):
    ops.zero_(z)
    ops.affine_act(x, y, scale=sc, shift=sh)
    ops.conv_gemm(a, w, c, B=M, Hs=1, Ws=1, Cs=K, Hd=1, Wd=1, Cd=N)
  torch.cuda.synchronize()
  print('known bytes: zero W=%d; affine_act R=%d W=%d; conv_gemm R>=%d W=%d' % (z.numel() * 4, x.numel() * 2, x.numel() * 2, (M * K + K * N) * 2, M * N * 2))


if __name__ == '__main__':
  main()
