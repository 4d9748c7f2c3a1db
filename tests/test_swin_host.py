"""Host-side index tables of the Video-Swin branch (carla_garage_amd/swin.py) against the tensor operations they replace
(team_code/video_swin_transformer.py:40-69 window_partition / window_reverse, :233-260 pad + roll, :305-309 PatchMerging slices,
:329-342 compute_mask), restated with torch on the CPU."""
import pytest
import torch
import torch.nn.functional as F

from carla_garage_amd.swin import merge_maps, window_geometry, window_maps


def _partition(x, ws):
  b, d, h, w, c = x.shape
  x = x.view(b, d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2], c)
  return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], c)


@pytest.mark.parametrize('dims', [(3, 16, 16), (3, 8, 8), (3, 64, 64), (2, 9, 11)])
@pytest.mark.parametrize('shifted', [False, True])
def test_window_maps_equal_pad_roll_partition_and_back(dims, shifted):
  B, C = 2, 4
  D, H, W = dims
  ws, ss = window_geometry(dims, (8, 7, 7), (4, 3, 3) if shifted else (0, 0, 0))
  assert ws[0] == D and ss[0] == 0  # the time axis (3 frames) is never larger than the window: one window, no shift along it
  x = torch.randn(B, D, H, W, C)
  fwd, rev, mask, nW, n = window_maps(B, D, H, W, ws, ss)
  xp = F.pad(x, (0, 0, 0, (ws[2] - W % ws[2]) % ws[2], 0, (ws[1] - H % ws[1]) % ws[1], 0, (ws[0] - D % ws[0]) % ws[0]))
  xs = torch.roll(xp, shifts=tuple(-s for s in ss), dims=(1, 2, 3)) if any(ss) else xp
  want = _partition(xs, ws)
  rows = x.reshape(-1, C)
  got = torch.where(fwd[:-8, None] >= 0, rows[fwd[:-8].clamp(min=0).long()], torch.zeros(1))
  assert (fwd[-8:] == -1).all() and torch.equal(got.view_as(want), want)
  y = torch.randn_like(want)
  b, d, h, w, _ = xs.shape
  yr = y.view(b, d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2], C).permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(b, d, h, w, C)
  yr = (torch.roll(yr, shifts=ss, dims=(1, 2, 3)) if any(ss) else yr)[:, :D, :H, :W].contiguous()
  assert torch.equal(y.reshape(-1, C)[rev.long()].view_as(yr), yr)
  if any(ss):
    img = torch.zeros((1, d, h, w, 1))
    cnt = 0
    for sd in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
      for sh in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
        for sw in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
          img[:, sd, sh, sw, :] = cnt
          cnt += 1
    mw = _partition(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    am = am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)
    assert torch.equal(mask, am)
  else:
    assert mask is None


@pytest.mark.parametrize('hw', [(64, 64), (7, 9)])
def test_merge_maps_equal_the_strided_slices(hw):
  B, D, C = 2, 3, 4
  H, W = hw
  x = torch.randn(B, D, H, W, C)
  xp = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
  want = torch.cat([xp[:, :, 0::2, 0::2], xp[:, :, 1::2, 0::2], xp[:, :, 0::2, 1::2], xp[:, :, 1::2, 1::2]], -1).reshape(-1, 4 * C)
  maps, h2, w2 = merge_maps(B, D, H, W)
  rows = x.reshape(-1, C)
  got = torch.cat([torch.where(m[:, None] >= 0, rows[m.clamp(min=0).long()], torch.zeros(1)) for m in maps], -1)
  assert (h2, w2) == ((H + 1) // 2, (W + 1) // 2) and torch.equal(got, want)
