import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle import tfpp_port as P
from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.model import LidarCenterNet
from carla_garage_amd.trainer import Trainer
from carla_garage_amd.graph import GraphedTrainStep
batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
  batch[k] = v.cuda()
dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
def run(single, graph):
  for k in ('TFPP_BRANCH_STREAMS', 'TFPP_SIDE_STREAM'):
    if single: os.environ[k] = '0'
    else: os.environ.pop(k, None)
  m = LidarCenterNet(GlobalConfig(tfpp_dtype=dtype)); m.load_state_dict(P.make_state_dict(), strict=True); m = m.cuda().train()
  for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
  m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0
  tr = Trainer(m, lr=1e-5)
  v1 = tr.train_step(batch).detach().float().cpu().numpy().copy()
  g1 = tr.eng.flat_grad.detach().double().cpu().numpy().copy()
  if graph:
    gs = GraphedTrainStep(tr, batch, warmup=0)
    v2 = gs(batch).detach().float().cpu().numpy().copy()
  else:
    v2 = tr.train_step(batch).detach().float().cpu().numpy().copy()
  torch.cuda.synchronize()
  g2 = tr.eng.flat_grad.detach().double().cpu().numpy().copy()
  return v1, g1, v2, g2
ref = run(True, False)
for name, args in (('single again', (True, False)), ('lanes', (False, False)), ('lanes+graph', (False, True))):
  r = run(*args)
  print(name, 'step1 loss rel', np.max(np.abs(r[0] - ref[0]) / np.abs(ref[0])), 'grad rel', np.linalg.norm(r[1] - ref[1]) / np.linalg.norm(ref[1]),
        '| step2 loss rel', np.max(np.abs(r[2] - ref[2]) / np.abs(ref[2])), 'grad rel', np.linalg.norm(r[3] - ref[3]) / np.linalg.norm(ref[3]), flush=True)
