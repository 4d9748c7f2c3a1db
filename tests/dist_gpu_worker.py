"""Worker of tests/test_dist_gpu.py: runs in its own process with a ONE-rank 'nccl' (= RCCL) process group and
TFPP_FORCE_COLLECTIVES=1, so the whole data-parallel step of team_code/train.py:516-520,898 -- two backward segments, the
asynchronous all-reduce of the early-finishing slice between them, the all-reduce of the rest, the optimizer waiting on both,
and the same thing as two hipGraphs with the RCCL call between the replays -- executes on a 1-GPU box.  Prints one JSON line."""
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

  # reference: the local single-segment step (no collective)
  tr0 = Trainer(fresh(), lr=1e-5)
  tr0.exchange = False
  assert not tr0.overlap_enabled()
  v0 = tr0.train_step(batch).float().cpu().numpy()
  g0 = tr0.eng.flat_grad.detach().double().cpu().numpy()
  p0 = tr0.flat_param.detach().double().cpu().numpy()
  n0 = dict(calls)

  # eager step with the exchange: two segments + async all-reduce in between + all-reduce of the head
  tr1 = Trainer(fresh(), lr=1e-5)
  assert tr1.overlap_enabled() and tr1.world == 1
  v1 = tr1.train_step(batch).float().cpu().numpy()
  torch.cuda.synchronize()
  g1 = tr1.eng.flat_grad.detach().double().cpu().numpy()
  p1 = tr1.flat_param.detach().double().cpu().numpy()
  n1 = dict(calls)

  # the same as two hipGraphs with the RCCL calls between / after the replays (second step of tr1 vs second local step of tr0)
  v0b = tr0.train_step(batch).float().cpu().numpy()
  g0b = tr0.eng.flat_grad.detach().double().cpu().numpy()
  gs = GraphedTrainStep(tr1, batch, warmup=0)
  assert gs.split and gs.graph2 is not None
  before = dict(calls)
  v2 = gs(batch).float().cpu().numpy()
  torch.cuda.synchronize()
  g2 = tr1.eng.flat_grad.detach().double().cpu().numpy()
  after = dict(calls)
  rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))
  out = {'world': dist.get_world_size(), 'backend': dist.get_backend(),
         'calls_local': n0['all_reduce'], 'calls_eager_step': n1['all_reduce'] - n0['all_reduce'], 'async_eager_step': n1['async'] - n0['async'],
         'bytes_eager_step': n1['bytes'] - n0['bytes'], 'arena_bytes': int(tr1.eng.flat_grad.numel()) * 4,
         'calls_graph_step': after['all_reduce'] - before['all_reduce'], 'async_graph_step': after['async'] - before['async'],
         'loss_eager': float(np.max(np.abs(v1 - v0) / np.abs(v0))), 'grad_eager': rel(g1, g0), 'param_eager': rel(p1, p0),
         'loss_graph': float(np.max(np.abs(v2 - v0b) / np.abs(v0b))), 'grad_graph': rel(g2, g0b)}
  print('RESULT ' + json.dumps(out), flush=True)
  torch.cuda.synchronize()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
