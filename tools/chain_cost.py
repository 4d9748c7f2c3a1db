"""Timing diagnostic (results of the patched steps are garbage by construction): marginal cost of each kernel family on the critical path
of the captured training step = step time with everything - step time with that family's launches removed from the capture.  The step
is chain-bound, so per-kernel durations from a profiler overstate what most families cost; this measures it."""
import json
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from carla_garage_amd._lib import lib  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402

GROUPS = {
    'bn_apply_act(fwd)': ['tfpp_affine_act'],
    'bn_finalize': ['tfpp_bn_finalize_partials', 'tfpp_bn_finalize'],
    'bn_bwd': ['tfpp_bn_bwd_reduce', 'tfpp_bn_bwd_apply', 'tfpp_bn_bwd_apply_rows'],
    'se': ['tfpp_se_gate_fwd', 'tfpp_se_gate_bwd', 'tfpp_se_dgate', 'tfpp_se_bwd_apply_bns', 'tfpp_mean_hw'],
    'residual_dropout+layernorm': ['tfpp_add_dropout', 'tfpp_layernorm_fwd', 'tfpp_layernorm_bwd'],
    'colsum': ['tfpp_colsum'],
    'act_bwd': ['tfpp_act_bwd'],
    'wgrad': ['tfpp_conv_wgrad', 'tfpp_conv_wgrad_stage', 'tfpp_conv_wgrad_batch'],
    'attention': ['tfpp_attn_fwd', 'tfpp_attn_bwd'],
    'bilinear': ['tfpp_bilinear_fwd', 'tfpp_bilinear_bwd'],
    'optimizer+repack': ['tfpp_adamw_amsgrad', 'tfpp_pack_multi'],
    'copies': ['tfpp_copy_rows', 'tfpp_axpy', 'tfpp_zero', 'tfpp_pack2d'],
    'bgemm': ['tfpp_bgemm'],
    'pool+boundary': ['tfpp_avgpool_fwd', 'tfpp_avgpool_bwd_add', 'tfpp_nchw_to_nhwc_affine', 'tfpp_nhwc_to_nchw', 'tfpp_nchw_to_nhwc_pad', 'tfpp_add_bcast', 'tfpp_mul_pixmask'],
    'heads_small': ['tfpp_gru_fwd', 'tfpp_gru_bwd', 'tfpp_softmax_fwd', 'tfpp_softmax_bwd', 'tfpp_bn1d_scalar', 'tfpp_sum_f32'],
    'losses': ['tfpp_ce_loss', 'tfpp_reg_loss'],
    'conv_gemm(all fwd+dgrad)': ['tfpp_conv_gemm'],
}


def main(bs=12, steps=20):
  cfg = GlobalConfig(tfpp_dtype='bf16')
  dev = torch.device('cuda:0')
  torch.manual_seed(0)
  tr = Trainer(LidarCenterNet(cfg).to(dev).train(), lr=1e-5)
  batch = bench.synthetic_batch(bs, cfg, dev, 1234)
  tr.train_step(batch)
  lib.load()

  def timed():
    step = GraphedTrainStep(tr, batch, warmup=1)
    for _ in range(3):
      step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

  base = timed()
  res = {'all': round(base, 3)}
  names = sys.argv[1:] or list(GROUPS)
  for g in names:
    fns = [f for f in GROUPS[g] if f in lib._fns]
    for f in fns:
      lib.__dict__[f] = lambda *a: None
    try:
      res[g] = round(base - timed(), 3)
    except Exception as e:  # pylint: disable=broad-except
      res[g] = f'{type(e).__name__}: {e}'
    for f in fns:
      lib.__dict__.pop(f, None)
    print(g, res[g], flush=True)
  res['all_again'] = round(timed(), 3)
  print(json.dumps(res))


if __name__ == '__main__':
  main()
