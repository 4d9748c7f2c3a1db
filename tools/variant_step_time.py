"""Training step (bs = 12, bf16, hipGraph replay) and the bs = 1 eval forward call of the non-default planning-head variants next to the default
configuration, in one process on one box:  python tools/variant_step_time.py [steps]  ->  one JSON line per configuration.

default        use_controller_input_prediction = 1 (checkpoints + target speed)          -- BASELINE config 3, the bench line
multi_wp       use_wp_gru = 1, use_controller_input_prediction = 0, multi_wp_output = 1   -- two waypoint hypotheses + path-selection logit
tp_attention   tp_attention = 1                                                           -- target-point token, attention-returning decoder
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import synthetic_batch  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.graph import GraphedTrainStep  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402

VARIANTS = {
    'default': {},
    'multi_wp': dict(use_wp_gru=True, use_controller_input_prediction=False, multi_wp_output=True),
    'tp_attention': dict(tp_attention=True),
}


def main():
  steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
  dev = torch.device('cuda:0')
  for name, over in VARIANTS.items():
    cfg = GlobalConfig(tfpp_dtype='bf16', **over)
    torch.manual_seed(0)
    model = LidarCenterNet(cfg).to(dev).train()
    batch = synthetic_batch(12, cfg, dev, 1234)
    tr = Trainer(model, lr=cfg.lr)
    for _ in range(3):  # eager steps: the arenas move into the observed completion order
      tr.train_step(batch)
    step = GraphedTrainStep(tr, batch, warmup=1)
    for _ in range(3):
      vals = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      vals = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    out = {'variant': name, 'losses': len(tr.loss_names), 'train_ms_per_step_bs12_bf16': round(ms, 3), 'samples_per_s': round(12e3 / ms, 1),
           'weighted_loss': round(float(tr.total_loss(vals)), 5)}
    del step
    model.eval()
    model.eval_graph_after = 2  # (TFPP_EVAL_GRAPH_AFTER=2: the module captures its own eval forward)
    one = synthetic_batch(1, cfg, dev, 99)
    inp = [one[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
    with torch.inference_mode():
      for _ in range(5):  # two eager calls, the capture, two replays
        o = model(*inp)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        o = model(*inp)
        torch.cuda.synchronize()
      out['forward_call_tick_ms_bs1_bf16'] = round(1e3 * (time.perf_counter() - t0) / steps, 3)
    if o[7] is not None:
      out['attention_weights'] = [round(v, 4) for v in o[7]]
    print(json.dumps(out), flush=True)
    del tr, model
    torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
