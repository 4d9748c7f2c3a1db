"""Restatement of ``timm==0.6.7`` ``create_model('regnety_032', features_only=True)``.

Test infrastructure (see oracle/__init__.py).  timm is an un-vendored dependency
of the reference (team_code/requirements.txt:163; call sites
team_code/transfuser.py:25,52-55, team_code/aim.py:19) and is not installed, so
its published RegNet-Y algorithm (Radosavovic et al., "Designing Network Design
Spaces") is restated here in the shape timm exposes it: a ``ModuleDict`` with
children ``stem, s1..s4``, ``return_layers`` and ``feature_info.info``, which is
the consumer-side contract the reference relies on (team_code/transfuser.py:63-66,
96-99,159-172,207-220; parameter-name patterns team_code/model.py:586-589).

Parity of this restatement is "unpinned" against timm itself (no timm, no
checkpoints); its block arithmetic is pinned bit-exactly against the
independently written HF ``transformers`` RegNet-Y in tests/test_oracle.py.

RegNetY-3.2GF: w0=80 wa=42.63 wm=2.66 depth=21 group_size=24 se_ratio=0.25
-> widths [72,216,576,1512], depths [2,5,13,1], stem width 32.
"""
import sys
import types
import torch
from torch import nn

WIDTHS = (72, 216, 576, 1512)
DEPTHS = (2, 5, 13, 1)
GROUP_W = 24
STEM_W = 32
SE_RATIO = 0.25


class _ConvBn(nn.Module):
  """conv (no bias) -> BatchNorm2d -> optional ReLU; children named ``conv`` / ``bn`` like timm's ConvNormAct."""

  def __init__(self, cin, cout, k, stride=1, groups=1, act=True):
    super().__init__()
    self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)
    self.bn = nn.BatchNorm2d(cout)
    self.act = act

  def forward(self, x):
    x = self.bn(self.conv(x))
    return torch.relu(x) if self.act else x


class _SE(nn.Module):
  """squeeze-excite: mean(H,W) -> fc1 -> ReLU -> fc2 -> sigmoid -> scale."""

  def __init__(self, chs, rd):
    super().__init__()
    self.fc1 = nn.Conv2d(chs, rd, 1, bias=True)
    self.fc2 = nn.Conv2d(rd, chs, 1, bias=True)

  def forward(self, x):
    s = x.mean((2, 3), keepdim=True)
    s = self.fc2(torch.relu(self.fc1(s)))
    return x * torch.sigmoid(s)


class _Bottleneck(nn.Module):
  """RegNet-Y bottleneck (bottle_ratio 1): 1x1 -> grouped 3x3 (stride) -> SE -> 1x1, + shortcut, ReLU."""

  def __init__(self, cin, cout, stride):
    super().__init__()
    self.conv1 = _ConvBn(cin, cout, 1)
    self.conv2 = _ConvBn(cout, cout, 3, stride, groups=cout // GROUP_W)
    self.se = _SE(cout, int(round(cin * SE_RATIO)))
    self.conv3 = _ConvBn(cout, cout, 1, act=False)
    self.downsample = _ConvBn(cin, cout, 1, stride, act=False) if (cin != cout or stride != 1) else None
    nn.init.zeros_(self.conv3.bn.weight)  # timm zero_init_last

  def forward(self, x):
    sc = x if self.downsample is None else self.downsample(x)
    y = self.conv3(self.se(self.conv2(self.conv1(x))))
    return torch.relu(y + sc)


class _FeatureInfo:

  def __init__(self, info):
    self.info = info

  def channels(self):
    return [i['num_chs'] for i in self.info]


class RegNetYFeatures(nn.ModuleDict):
  """timm FeatureListNet look-alike: iterate ``.items()`` in order stem, s1..s4."""

  def __init__(self, in_chans=3):
    super().__init__()
    self['stem'] = _ConvBn(in_chans, STEM_W, 3, 2)
    cin = STEM_W
    for i, (w, d) in enumerate(zip(WIDTHS, DEPTHS)):
      stage = nn.Sequential()
      for k in range(d):
        stage.add_module(f'b{k + 1}', _Bottleneck(cin, w, 2 if k == 0 else 1))
        cin = w
      self[f's{i + 1}'] = stage
    self.return_layers = {n: str(i) for i, n in enumerate(['stem', 's1', 's2', 's3', 's4'])}
    self.feature_info = _FeatureInfo([dict(num_chs=STEM_W, reduction=2, module='stem')] + [
        dict(num_chs=w, reduction=4 * 2**i, module=f's{i + 1}') for i, w in enumerate(WIDTHS)
    ])
    for m in self.modules():
      if isinstance(m, nn.Conv2d) and m.bias is None:
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        nn.init.normal_(m.weight, 0.0, (2.0 / fan_out)**0.5)

  def forward(self, x):
    outs = []
    for _, m in self.items():
      x = m(x)
      outs.append(x)
    return outs


def create_model(name, pretrained=False, features_only=True, in_chans=3, **_):
  """``timm.create_model`` stand-in.  ``pretrained`` is ignored (no network)."""
  if name != 'regnety_032' or not features_only:
    raise ValueError(f'oracle timm restatement only provides regnety_032 features_only, got {name}')
  return RegNetYFeatures(in_chans)


class DropPath(nn.Module):
  """timm.models.layers.DropPath (stochastic depth); identity in eval."""

  def __init__(self, drop_prob=0.0):
    super().__init__()
    self.drop_prob = drop_prob

  def forward(self, x):
    if self.drop_prob == 0.0 or not self.training:
      return x
    keep = 1.0 - self.drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    return x * mask / keep


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
  return nn.init.trunc_normal_(tensor, mean, std, a, b)


def install_as_timm():
  """Register this module as ``timm`` (+ ``timm.models.layers``) in ``sys.modules``."""
  timm = types.ModuleType('timm')
  timm.create_model = create_model
  models = types.ModuleType('timm.models')
  layers = types.ModuleType('timm.models.layers')
  layers.DropPath = DropPath
  layers.trunc_normal_ = trunc_normal_
  models.layers = layers
  timm.models = models
  sys.modules['timm'] = timm
  sys.modules['timm.models'] = models
  sys.modules['timm.models.layers'] = layers
  return timm
