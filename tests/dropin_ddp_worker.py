"""Worker of tests/test_dropin_gpu.py::test_under_distributed_data_parallel_one_rank_rccl: its own process, ONE-rank 'nccl' (= RCCL) group,
TFPP_FORCE_COLLECTIVES=1.  The module is wrapped in torch's DistributedDataParallel with the arguments of team_code/train.py:516-520 and
driven by the restated train.py loop (tests/test_dropin_gpu.py::train_py_loop) with the fused optimizer; a Trainer on the same weights and
batches (no collectives) is the reference.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['TFPP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  import test_dropin_gpu as T
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  torch.cuda.set_device(0)
  lr = 1e-4
  batches = T._batches(5)
  os.environ['TFPP_FORCE_COLLECTIVES'] = '0'
  want, want_param, _ = T._trainer_reference(batches, lr)   # before the process group exists: a purely local Trainer
  os.environ['TFPP_FORCE_COLLECTIVES'] = '1'
  dist.init_process_group('nccl', init_method='env://', rank=0, world_size=1)
  calls = {'n': 0, 'bytes': 0}
  real = dist.all_reduce

  def counting(t, *a, **k):
    if t.numel() <= 2:  # Trainer.agree_on_layout: one tiny MAX all-reduce per arena layout, not part of the gradient exchange
      return real(t, *a, **k)
    calls['n'] += 1
    calls['bytes'] += t.numel() * t.element_size()
    return real(t, *a, **k)

  dist.all_reduce = counting
  m = T._model()
  ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=None, output_device=None, broadcast_buffers=False, find_unused_parameters=False)
  managed = [n for n, p in m.named_parameters() if p.requires_grad and n not in ddp.parameters_to_ignore and '.' + n not in ddp.parameters_to_ignore]
  opt = FlatAdamW(ddp.parameters(), lr=lr, amsgrad=True)
  w = normalized_loss_weights(m.config)
  got, per_step, per_bytes, arena_at = [], [], [], []
  for b in batches:
    n0, b0 = calls['n'], calls['bytes']
    got += T.train_py_loop(m, opt, [b], w, wrapper=ddp)
    torch.cuda.synchronize()
    per_step.append(calls['n'] - n0)
    per_bytes.append(calls['bytes'] - b0)
    arena_at.append(int(m.__dict__['_dropin_step'].eng.flat_grad.numel()) * 4)
  step = m.__dict__['_dropin_step']
  plan = next(iter(step.plans.values()))
  got_param = T._params_by_name(m)
  out = {'ddp_params': len(managed), 'ignored': len(ddp.parameters_to_ignore), 'trainable': len([p for p in m.parameters() if p.requires_grad]),
         'calls_per_step': per_step, 'arena_bytes_per_step': per_bytes, 'arena_bytes_at_step': arena_at, 'arena_bytes': int(step.eng.flat_grad.numel()) * 4,
         'buckets': len(step.eng.buckets.ranges()), 'early_signals': len(plan.program[1]) if plan.program else -1, 'wait_timeouts': step.eng.buckets.timed_out(),
         'graph_steps': plan.count - 3 if plan.B1 is not None else 0, 'lr': lr,
         'loss_rel': [abs(a - b) / abs(b) for a, b in zip(got, want)], 'param_abs': T._check_params(got_param, want_param, 5, lr)[1], 'param_rel': T._check_params(got_param, want_param, 5, lr)[0]}
  print('RESULT ' + json.dumps(out), flush=True)
  torch.cuda.synchronize()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
