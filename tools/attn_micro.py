#!/usr/bin/env python
"""Micro-benchmark of the fused attention kernels against the unfused bgemm / softmax path on the four fusion scales (bs = 12).
Launches are replayed from a hipGraph, so the numbers are GPU time."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402


def timed(fn, iters=10):
  fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(iters):
      fn()
  g.replay()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  g.replay()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


def main():
  B, nh, T = 12, 4, 320
  dt = torch.bfloat16
  for d_real, dp in ((18, 24), (54, 56), (144, 144), (378, 384)):
    npk = 3 * nh * dp
    qkv = (torch.randn(B * T, npk, device='cuda') * 0.5).to(dt)
    q, k, v = qkv.view(-1)[0:], qkv.view(-1)[nh * dp:], qkv.view(-1)[2 * nh * dp:]
    O = torch.empty(B, T, nh * dp, device='cuda', dtype=dt)
    dO = torch.randn(B, T, nh * dp, device='cuda').to(dt)
    lse = torch.empty(B * nh * T, device='cuda')
    delta = torch.empty_like(lse)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv.view(-1)[0:], dqkv.view(-1)[nh * dp:], dqkv.view(-1)[2 * nh * dp:]
    geo = dict(B=B, nh=nh, T=T, d=dp, ld_q=npk, ld_kv=npk, ld_o=nh * dp, scale=1.0 / math.sqrt(d_real))
    S = torch.empty(B, nh, T, T, device='cuda', dtype=dt)
    dP = torch.empty_like(S)

    def unfused_fwd():
      ops.bgemm(q, k, S, M=T, N=T, K=dp, lda=npk, ldb=npk, ldc=T, batch0=B, batch1=nh, a_bs=(T * npk, dp), b_bs=(T * npk, dp), c_bs=(nh * T * T, T * T))
      ops.softmax_fwd(S, B * nh * T, T, T, alpha=geo['scale'])
      ops.bgemm(S, v, O, M=T, N=dp, K=T, lda=T, ldb=npk, ldc=nh * dp, batch0=B, batch1=nh, a_bs=(nh * T * T, T * T), b_bs=(T * npk, dp),
                c_bs=(T * nh * dp, dp), b_km=True)

    def unfused_bwd():
      ops.bgemm(S, dO, dv, M=T, N=dp, K=T, lda=T, ldb=nh * dp, ldc=npk, batch0=B, batch1=nh, a_bs=(nh * T * T, T * T), b_bs=(T * nh * dp, dp),
                c_bs=(T * npk, dp), a_km=True, b_km=True)
      ops.bgemm(dO, v, dP, M=T, N=T, K=dp, lda=nh * dp, ldb=npk, ldc=T, batch0=B, batch1=nh, a_bs=(T * nh * dp, dp), b_bs=(T * npk, dp),
                c_bs=(nh * T * T, T * T))
      ops.softmax_bwd(S, dP, B * nh * T, T, T, alpha=geo['scale'])
      ops.bgemm(dP, k, dq, M=T, N=dp, K=T, lda=T, ldb=npk, ldc=npk, batch0=B, batch1=nh, a_bs=(nh * T * T, T * T), b_bs=(T * npk, dp),
                c_bs=(T * npk, dp), b_km=True)
      ops.bgemm(dP, q, dk, M=T, N=dp, K=T, lda=T, ldb=npk, ldc=npk, batch0=B, batch1=nh, a_bs=(nh * T * T, T * T), b_bs=(T * npk, dp),
                c_bs=(T * npk, dp), a_km=True, b_km=True)

    if '--trace' in sys.argv:  # phase stamps of workgroup (0, 0): 100 MHz ticks since kernel start
      tr = torch.zeros(8, device='cuda')
      p = ops._attn_params(q, k, v, O, lse, delta=tr, **geo)
      import ctypes
      from carla_garage_amd._lib import lib
      lib.tfpp_attn_fwd(ctypes.byref(p), 1, ops.stream())
      torch.cuda.synchronize()
      names = ('scores', 'softmax', 'sync', 'V staged', 'PV chunk0', 'O chunk0 stored', 'end')
      print('  trace d=%d: ' % d_real + ', '.join('%s %.1f us' % (n, t / 100.0) for n, t in zip(names, tr.cpu().tolist())))
    f_fwd = timed(lambda: ops.attn_fwd(q, k, v, O, lse, **geo))
    f_bwd = timed(lambda: ops.attn_bwd(q, k, v, O, lse, dO, dq, dk, dv, delta, **geo))
    u_fwd = timed(unfused_fwd)
    u_bwd = timed(unfused_bwd)
    print(f'd={d_real:3d} (store {dp:3d})  fused fwd {f_fwd:7.1f} us  bwd {f_bwd:7.1f} us   unfused fwd {u_fwd:7.1f} us  bwd {u_bwd:7.1f} us', flush=True)


if __name__ == '__main__':
  main()
