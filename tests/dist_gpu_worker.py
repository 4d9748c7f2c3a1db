"""Worker of tests/test_dist_gpu.py: runs in its own process with a ONE-rank 'nccl' (= RCCL) process group and
TFPP_FORCE_COLLECTIVES=1, so the whole data-parallel step of team_code/train.py:516-520,898 -- one backward pass that raises a completion
signal per gradient bucket, one asynchronous RCCL all-reduce per bucket behind its signal, the optimizer bucket by bucket as they land,
and the same thing with the pass replayed from ONE hipGraph -- executes on a 1-GPU box.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['TFPP_FORCE_COLLECTIVES'] = '1'
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
  from oracle import tfpp_port as P  # deterministic weights / inputs only
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  from carla_garage_amd.trainer import Trainer
  from carla_garage_amd.graph import GraphedTrainStep
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', init_method='env://', rank=0, world_size=1)
  calls = {'all_reduce': 0, 'async': 0, 'bytes': 0}
  real = dist.all_reduce

  def counting(t, *a, **k):
    if t.numel() <= 2:  # the layout-agreement check of Trainer.agree_on_layout (one tiny MAX all-reduce per arena layout): not part of the exchange
      calls['agreement'] = calls.get('agreement', 0) + 1
      return real(t, *a, **k)
    calls['all_reduce'] += 1
    calls['async'] += int(bool(k.get('async_op', False)))
    calls['bytes'] += t.numel() * t.element_size()
    return real(t, *a, **k)

  dist.all_reduce = counting
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()

  def fresh():
    m = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32'))
    m.load_state_dict(P.make_state_dict(), strict=True)
    m.cuda().train()
    for mod in m.modules():
      if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
    m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0
    return m

  by_name = lambda tr: np.concatenate([tr.eng.grads[n].detach().double().cpu().numpy().ravel() for n, p in tr.model.named_parameters() if p.requires_grad])
  params = lambda tr: np.concatenate([p.detach().double().cpu().numpy().ravel() for p in tr.model.parameters()])

  # reference: the local step (no collective)
  tr0 = Trainer(fresh(), lr=1e-5)
  tr0.exchange = False
  assert not tr0.exchange_enabled()
  v0 = tr0.train_step(batch).float().cpu().numpy()
  g0, p0 = by_name(tr0), params(tr0)
  n0 = dict(calls)

  # eager step with the exchange: one all-reduce per bucket of the arena
  tr1 = Trainer(fresh(), lr=1e-5)
  assert tr1.exchange_enabled() and tr1.world == 1
  v1 = tr1.train_step(batch).float().cpu().numpy()
  torch.cuda.synchronize()
  g1, p1 = by_name(tr1), params(tr1)
  n1 = dict(calls)
  buckets_static, arena_static = len(tr1.eng.buckets.ranges()), int(tr1.eng.flat_grad.numel()) * 4

  # steps 2-3: the arenas move into the observed completion order; step 4 replayed from ONE hipGraph with the RCCL calls after the replay
  # call returns (they wait for the in-graph completion signals) -- against the fourth local step of tr0
  for _ in range(2):
    tr0.train_step(batch)
    tr1.train_step(batch)
  v0b = tr0.train_step(batch).float().cpu().numpy()
  g0b = by_name(tr0)
  gs = GraphedTrainStep(tr1, batch, warmup=0)
  assert tr1.step_count == 3 and tr1.layout_final
  before = dict(calls)
  v2 = gs(batch).float().cpu().numpy()
  torch.cuda.synchronize()
  g2 = by_name(tr1)
  after = dict(calls)
  rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
  out = {'world': dist.get_world_size(), 'backend': dist.get_backend(),
         'layout_agreements': calls.get('agreement', 0), 'calls_local': n0['all_reduce'], 'calls_eager_step': n1['all_reduce'] - n0['all_reduce'], 'async_eager_step': n1['async'] - n0['async'],
         'bytes_eager_step': n1['bytes'] - n0['bytes'], 'arena_bytes': arena_static, 'arena_bytes_observed': int(tr1.eng.flat_grad.numel()) * 4,
         'calls_graph_step': after['all_reduce'] - before['all_reduce'], 'async_graph_step': after['async'] - before['async'],
         'bytes_graph_step': after['bytes'] - before['bytes'], 'buckets_static': buckets_static, 'buckets_observed': len(tr1.eng.buckets.ranges()),
         'early_signals': len(gs.program[1]), 'poisoned': tr1.eng.buckets.poisoned, 'wait_timeouts': tr1.eng.buckets.timed_out(),
         'loss_eager': float(np.max(np.abs(v1 - v0) / np.abs(v0))), 'grad_eager': rel(g1, g0), 'param_eager': rel(p1, p0),
         'loss_graph': float(np.max(np.abs(v2 - v0b) / np.abs(v0b))), 'grad_graph': rel(g2, g0b)}
  print('RESULT ' + json.dumps(out), flush=True)
  torch.cuda.synchronize()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
