"""CenterNet heat-map decode + box conversion: oracle vs. the reference's golden output (CPU), HIP path
vs. both through the model boundary (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_port as D

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'centernet_decode.npz')
CASES = ('b2', 'few')


def _case(name):
  g = np.load(GOLDEN)
  batch, seed, peaks = (int(v) for v in g[f'{name}.args'])
  return D.make_maps(batch, seed, peaks), g[f'{name}.boxes'], g[f'{name}.carla']


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference_golden(name):
  maps, boxes, carla = _case(name)
  got = D.decode_heatmap(*maps)
  assert got.dtype == torch.float32 and np.array_equal(got.numpy(), boxes)
  mine = D.convert_features_to_bb_metric(maps)
  assert len(mine) == len(carla) and all(np.array_equal(a, b) for a, b in zip(mine, carla))
  assert np.all(np.diff(boxes[..., 8], axis=1) < 0)  # scores strictly descending: the fixtures have no ties


@pytest.mark.gpu
@pytest.mark.parametrize('name', CASES)
def test_hip_decode_bit_exact(name):
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  maps, boxes, carla = _case(name)
  m = LidarCenterNet(GlobalConfig())
  dev_maps = [t.cuda() for t in maps]
  got = m.head.get_bboxes(*dev_maps, None, None)
  assert got.shape == boxes.shape and got.dtype == torch.float32 and got.is_cuda
  assert np.array_equal(got.cpu().numpy(), boxes)
  mine = m.convert_features_to_bb_metric(dev_maps + [None, None])
  assert len(mine) == len(carla) and all(np.array_equal(a, b) for a, b in zip(mine, carla))


@pytest.mark.gpu
def test_hip_decode_properties_on_model_output():
  """On the model's own (random-weight) head output: scores descending, every pick is a 3x3 local maximum of its class plane,
  coordinates inside the image, and the decode equals the oracle wherever the scores are distinct."""
  from oracle import tfpp_port as P
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  m = LidarCenterNet(GlobalConfig())
  m.load_state_dict(P.make_state_dict(), strict=True)
  m.cuda().eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1)])
  bb = out[6]
  got = m.head.get_bboxes(*bb[:5], None, None).cpu()
  want = D.decode_heatmap(*[t.float().cpu() for t in bb[:5]])
  s = got[0, :, 8]
  assert got.shape == (1, 100, 9) and torch.all(s[:-1] >= s[1:])  # top-k, descending
  # every pick with a positive score is a 3x3 local maximum of its class plane and carries that pixel's score
  heat = bb[0].float().cpu()
  hmax = torch.nn.functional.max_pool2d(heat, 3, stride=1, padding=1)
  kept = (heat * (hmax == heat).float()).reshape(-1)
  assert torch.equal(torch.sort(kept, descending=True).values[:100], s)
  # and wherever the oracle's scores are distinct (no tie-break freedom) the whole row is identical
  ws = want[0, :, 8]
  distinct = torch.ones(100, dtype=torch.bool)
  distinct[1:] &= ws[1:] != ws[:-1]
  distinct[:-1] &= ws[:-1] != ws[1:]
  assert torch.equal(got[0][distinct], want[0][distinct])
