"""Eager bs=1 eval forwards of TransFuser++ for a rocprofv3 kernel trace (launch list of the 20 Hz tick, sensor_agent.py:456-461).
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/inf -- python tools/infer_trace.py [bf16|fp32] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg = GlobalConfig(tfpp_dtype=dtype)
torch.manual_seed(0)
model = LidarCenterNet(cfg).to(dev).eval()
b = synthetic_batch(1, cfg, dev, 99)
inp = [b[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
with torch.inference_mode():
  for _ in range(n):
    model(*inp)
    torch.cuda.synchronize()
print('done', n, flush=True)
