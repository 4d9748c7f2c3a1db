import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tfpp_port as P
from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.model import LidarCenterNet
from carla_garage_amd.trainer import Trainer
from carla_garage_amd.graph import GraphedTrainStep
from carla_garage_amd.engine import finishes_early
batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
  batch[k] = v.cuda()
def run(split):
  os.environ['TFPP_SPLIT_STEP'] = '1' if split else '0'
  m = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32')); m.load_state_dict(P.make_state_dict(), strict=True); m = m.cuda().train()
  for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
  m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0
  tr = Trainer(m, lr=1e-5)
  tr.train_step(batch)
  gs = GraphedTrainStep(tr, batch, warmup=0)
  gs(batch)
  torch.cuda.synchronize()
  return {n: g.detach().double().cpu().numpy().copy() for n, g in tr.eng.grads.items()}
a = run(False); b = run(True)
rows = []
for n in a:
  d = np.linalg.norm(a[n] - b[n]); r = d / (np.linalg.norm(a[n]) + 1e-30)
  rows.append((r, n, finishes_early(n), np.linalg.norm(a[n]), np.linalg.norm(b[n])))
rows.sort(reverse=True)
bad = [r for r in rows if r[0] > 1e-2]
print('params', len(rows), 'bad', len(bad), 'bad early', sum(1 for r in bad if r[2]), 'bad late', sum(1 for r in bad if not r[2]))
for r in rows[:25]:
  print('%.3e %-60s early=%s |a|=%.3e |b|=%.3e' % r)
