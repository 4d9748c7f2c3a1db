"""Squeeze-excite forward / backward launch chains alone on the chip: fused (tfpp_se_squeeze_gate / tfpp_se_bwd_squeeze) against the unfused
four-launch chains, per RegNet stage shape, HIP-event time over eager launches.  usage (GPU box): python tools/se_micro.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from carla_garage_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device('cuda:0')


def timed(fn):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / iters


for B in (1, 12):
  for (H, W, C, RD) in ((64, 256, 72, 8), (32, 128, 216, 54), (16, 64, 576, 144), (8, 32, 1512, 378), (16, 16, 576, 144)):
    x = torch.randn(B, H, W, C, device=dev).bfloat16()
    dy = torch.randn(B, H, W, C, device=dev).bfloat16()
    w1, b1 = torch.randn(RD, C, device=dev) * 0.05, torch.randn(RD, device=dev)
    w2, b2 = torch.randn(C, RD, device=dev) * 0.05, torch.randn(C, device=dev)
    g = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]

    def unfused_fwd():
      pool = ops.mean_hw(x)
      return (pool,) + ops.se_gate_fwd(pool, w1, b1, w2, b2)

    pool, hidden, gate = unfused_fwd()

    def unfused_bwd():
      dgate = ops.se_dgate(dy, x)
      return ops.se_gate_bwd(dgate, gate, hidden, pool, w1, w2, *g)

    def fused_bwd():
      dgate, dz1, dpool = ops.se_bwd_squeeze(dy, x, gate, hidden, w1, w2)
      return dpool

    print(f'B={B:2d} HW={H * W:6d} C={C:5d} RD={RD:4d}:  fwd unfused {timed(unfused_fwd):7.1f} us  fused {timed(lambda: ops.se_squeeze_gate(x, w1, b1, w2, b2)):7.1f} us'
          f'   bwd unfused {timed(unfused_bwd):7.1f} us  fused {timed(fused_bwd):7.1f} us', flush=True)
