"""Worker of tests/test_boundary_gpu.py: ONE-rank 'nccl' (= RCCL) group, TFPP_FORCE_COLLECTIVES=1.
  1. torch.distributed.optim.ZeroRedundancyOptimizer(params, optimizer_class=AdamW, lr, amsgrad=True) -- the reference's default
     (team_code/config.py:185, train.py:527-529) -- around the arena-backed parameters, against plain torch.optim.AdamW;
  2. DistributedDataParallel (train.py:516-520) around a module with the learnable loss weights of --learn_multi_task_weights
     (train.py:479-483): DDP manages the anchor and the ten weights, the arena is exchanged by the package; against the same loop without DDP.
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['TFPP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LOSSES = ('loss_target_speed', 'loss_checkpoint', 'loss_semantic', 'loss_bev_semantic', 'loss_depth', 'loss_center_heatmap', 'loss_wh', 'loss_offset',
          'loss_yaw_class', 'loss_yaw_res')


def main():
  import test_dropin_gpu as TD
  import test_boundary_gpu as TB
  from carla_garage_amd.losses import normalized_loss_weights
  from torch.distributed.optim import ZeroRedundancyOptimizer
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', init_method='env://', rank=0, world_size=1)
  lr = 1e-4
  batches = TD._batches(3)
  out = {}

  # ---- 1. ZeRO-1 wrapper of the reference's optimizer
  res = {}
  for kind in ('plain', 'zero'):
    m = TD._model()
    w = normalized_loss_weights(m.config)
    p0 = TB._params(m)
    if kind == 'zero':
      opt = ZeroRedundancyOptimizer(m.parameters(), optimizer_class=torch.optim.AdamW, lr=lr, amsgrad=True)   # train.py:527-529
    else:
      opt = torch.optim.AdamW(m.parameters(), lr=lr, amsgrad=True)
    res[kind] = (TD.train_py_loop(m, opt, batches, w), TB._params(m) - p0)
  out['zero_loss_rel'] = [abs(a - b) / abs(b) for a, b in zip(res['zero'][0], res['plain'][0])]
  d0, d1 = res['plain'][1].double(), res['zero'][1].double()
  out['zero_param_rel'] = float((d1 - d0).norm() / d0.norm())

  # ---- 2. DDP + learnable loss weights
  res = {}
  for kind in ('plain', 'ddp'):
    m = TD._model()
    learn = {}
    for i, k in enumerate(LOSSES):
      p = torch.nn.Parameter(torch.tensor(0.1 * i, dtype=torch.float32))
      m.register_parameter(name='weight_' + k, param=p)            # train.py:481-482
      learn[k] = p
    m.cuda()
    net = m
    if kind == 'ddp':
      net = torch.nn.parallel.DistributedDataParallel(m, device_ids=None, output_device=None, broadcast_buffers=False, find_unused_parameters=False)
      out['ddp_learn_managed'] = len([n for n, p in m.named_parameters() if p.requires_grad and n not in net.parameters_to_ignore and
                                      '.' + n not in net.parameters_to_ignore])
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, amsgrad=True)
    w0 = {k: float(v) for k, v in learn.items()}

    class Wrapped:  # TB._loop calls m(...) and m.compute_loss / m.parameters: forward through DDP, the rest on the module
      def __call__(self, **kw):
        return net(**kw)

      def compute_loss(self, **kw):
        return m.compute_loss(**kw)

      def parameters(self):
        return m.parameters()

    totals = TB._loop(Wrapped(), opt, batches, None, learn=learn)
    res[kind] = (totals, sum(int(abs(float(v) - w0[k]) > 0.5e-3) for k, v in learn.items()))
  out['ddp_learn_loss_rel'] = [abs(a - b) / abs(b) for a, b in zip(res['ddp'][0], res['plain'][0])]
  out['ddp_learn_weight_moves'] = res['ddp'][1]
  print('RESULT ' + json.dumps(out), flush=True)
  torch.cuda.synchronize()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
