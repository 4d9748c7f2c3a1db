"""Host-side dispatch logic of the C ABI (no kernel launches, runs without a GPU): which kernel / how many slices the library
plans for the shapes of the TransFuser++ step, and the invariants the engine relies on."""
import ctypes

import pytest

from carla_garage_amd._lib import lib, ConvParams, WgradParams, PackDesc, BF16, F32

FAKE = 0x10000  # an aligned, never dereferenced "device pointer"


@pytest.fixture(scope='module')
def L():
  lib.load()
  return lib


def conv(B, H, W, Cin, Cout, k=1, stride=1, G=1, mode=0, stats=False, ws=True):
  p = ConvParams()
  p.src = p.w = p.dst = FAKE
  pad = k // 2
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
  p.B, p.Hs, p.Ws, p.Cs, p.Hd, p.Wd, p.Cd = B, H, W, Cin, Ho, Wo, Cout
  p.R = p.S = k
  p.stride, p.pad, p.G, p.ks_g, p.n_g, p.mode = stride, pad, G, Cin // G, Cout // G, mode
  p.alpha, p.src_ld, p.dst_ld, p.res_ld = 1.0, Cin, Cout, Cout
  if stats:
    p.stats_partial, p.stats_rows = FAKE, 1
  if ws:
    p.splitk_ws, p.splitk_ws_floats = FAKE, 16 << 20
  return p


def wgrad(B, H, W, Cin, Cout, k=1, stride=1, G=1, ws=True):
  p = WgradParams()
  p.dy = p.x = p.dw = FAKE
  pad = k // 2
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
  p.B, p.Hs, p.Ws, p.Cs, p.Hd, p.Wd, p.Cd = B, H, W, Cin, Ho, Wo, Cout
  p.R = p.S = k
  p.stride, p.pad, p.G, p.ks_g, p.n_g, p.c_real = stride, pad, G, Cin // G, Cout // G, Cin // G
  p.x_ld, p.dy_ld, p.dw_ld = Cin, Cout, (Cin // G) * k * k
  if ws:
    p.ws, p.ws_floats = FAKE, 16 << 20
  return p


def variant(L, p, dt=BF16):
  return L.raw('tfpp_conv_gemm_variant')(ctypes.byref(p), dt)


def splits(L, p, dt=BF16):
  return L.raw('tfpp_conv_gemm_splits')(ctypes.byref(p), dt)


def wplan(L, p, dt=BF16):
  plan = (ctypes.c_int * 3)()
  assert L.raw('tfpp_conv_wgrad_stage')(ctypes.byref(p), dt, -1, plan, None) == 0
  return tuple(plan)


def test_conv_kernel_selection(L):
  # RegNet stage-3 1x1 conv of the image branch: LDS-DMA ring, 128x128 tiles from 256 tiles on
  assert variant(L, conv(12, 16, 64, 576, 576)) == 200
  assert variant(L, conv(12, 16, 16, 576, 576)) == 201          # LiDAR branch: too few tiles for 128x128
  assert variant(L, conv(3840, 1, 1, 1512, 6048)) == 202        # fusion MLP: K >= 1024 and >= 128 tiles of 256x128 (16 waves)
  assert variant(L, conv(3840, 1, 1, 576, 2304)) == 200         # K = 576: too short for the 144 KB ring
  assert variant(L, conv(12, 8, 32, 1512, 1512)) == 202         # image stage-4 1x1 conv
  assert variant(L, conv(12, 16, 64, 576, 576), F32) in (2, 3)  # fp32 never takes the bf16-only kernels
  assert variant(L, conv(12, 32, 128, 216, 216)) == 200           # stage-2 1x1 conv: K = 216 >= 200 runs the 64-deep LDS-DMA ring (K tail through the zero page)
  assert variant(L, conv(12, 64, 256, 72, 144)) in (0, 1, 2, 3, 4)  # K = 72 < 200: LDS-staged
  # 3x3 stride-1 with few channels per group: LDS-halo kernel (300 + 16-channel output fragments)
  assert variant(L, conv(12, 256, 1024, 32, 32, k=3)) == 302
  assert variant(L, conv(12, 256, 1024, 32, 8, k=3)) == 301
  assert variant(L, conv(12, 16, 64, 576, 576, k=3, G=24)) == 302
  assert variant(L, conv(12, 16, 64, 576, 576, k=3, G=24, mode=1)) == 302
  assert variant(L, conv(12, 32, 128, 216, 216, k=3, stride=2, G=9)) == 302  # stride 2, even maps: halo kernel with a 17 x 65 input halo
  assert variant(L, conv(12, 16, 64, 216, 216, k=3, stride=2, G=9, mode=1)) < 100  # its data gradient as built here (Hd = Hs/2) is not a stride-2 transpose
  assert variant(L, conv(12, 33, 128, 216, 216, k=3, stride=2, G=9)) < 100   # odd height: implicit GEMM
  assert variant(L, conv(12, 8, 8, 1512, 128, k=3)) >= 200 and variant(L, conv(12, 8, 8, 1512, 128, k=3)) < 300  # W < 32: no halo tiles


def test_split_k_only_for_few_tiles_and_long_reductions(L):
  assert splits(L, conv(132, 1, 1, 2048, 256), F32) > 1            # planning-head FFN: 12 tiles, K = 2048
  assert splits(L, conv(12, 8, 8, 1512, 128, k=3)) > 1             # K = 13608, 6 tiles
  assert splits(L, conv(12, 16, 64, 576, 576)) == 1                # fills the chip already
  assert splits(L, conv(132, 1, 1, 2048, 256, ws=False), F32) == 1  # no workspace, no split
  assert splits(L, conv(132, 1, 1, 2048, 256, stats=True), F32) == 1  # fused BN statistics need the whole reduction in one tile
  p = conv(132, 1, 1, 2048, 256)
  p.splitk_ws_floats = 132 * 256 * 3                                # room for 3 slices only
  assert 1 < splits(L, p, F32) <= 3


def test_stats_rows_are_the_tiles_of_the_selected_kernel(L):
  rows = lambda p, dt=BF16: L.raw('tfpp_conv_gemm_stats_rows')(ctypes.byref(p), dt)
  assert rows(conv(12, 16, 64, 576, 576, stats=True)) == 12 * 16 * 64 // 128          # LDS-DMA 128x128
  assert rows(conv(12, 16, 16, 576, 576, stats=True)) == 12 * 16 * 16 // 64           # LDS-DMA 64x128
  assert rows(conv(12, 16, 64, 576, 576, k=3, G=24, stats=True)) == 12 * 2 * 2        # halo: 8 x 32 pixel tiles
  assert rows(conv(12, 64, 256, 72, 72, stats=True)) == 12 * 64 * 256 // 128          # LDS-staged 128x32


def test_weight_gradient_plan(L):
  v, s, r = wplan(L, wgrad(12, 16, 64, 576, 576))
  assert v == 4 and s > 1 and s % 8 == 0 and r == 1                # 8-wave 128x128 LDS-DMA ring (8 GFLOP), whole XCD rounds of slices, slice sum
  v, s, r = wplan(L, wgrad(12, 16, 16, 576, 576))
  assert v == 2 and s > 1 and r == 1                               # LiDAR branch (2 GFLOP): 64x64 tiles, more slices
  assert wplan(L, wgrad(12, 32, 128, 216, 216))[0] == 2            # 216 channels fill 128-wide tiles to 71 % only: 64x64
  assert wplan(L, wgrad(12, 256, 1024, 32, 32, k=3))[0] == 3       # 3x3 halo
  assert wplan(L, wgrad(12, 16, 64, 576, 576, k=3, G=24))[0] == 3
  assert wplan(L, wgrad(12, 64, 64, 64, 64, k=3))[0] == 2          # 64-channel inputs on a small map: implicit GEMM wins
  assert wplan(L, wgrad(12, 64, 256, 72, 72), F32)[0] == 1         # fp32: LDS-staged 64x64
  assert wplan(L, wgrad(12, 128, 512, 8, 32, k=3, stride=2))[0] == 0  # stem: N <= 32, stride 2
  v, s, r = wplan(L, wgrad(3840, 1, 1, 1512, 6048))
  assert v == 4 and s == 1 and r == 0                              # 576 tiles of 128x128: single slice, added straight into the gradient
  v, s, r = wplan(L, wgrad(12, 16, 64, 576, 576, ws=False))
  assert r == 0 and s >= 1                                         # no workspace: atomics, no second stage
  p = wgrad(12, 16, 64, 576, 576)
  p.ws_floats = 576 * 576 * 3
  assert wplan(L, p)[1] <= 3                                       # slices shrink to what the workspace holds


def test_pack_plan_modes_and_workgroups(L):
  plan = L.raw('tfpp_pack_desc_plan')
  per = L.raw('tfpp_pack_elems_per_block')()

  def desc(kind, total, a, dtype=BF16):
    d = PackDesc()
    d.src = d.dst = FAKE
    d.total, d.kind, d.dtype = total, kind, dtype
    for i, v in enumerate(a):
      d.a[i] = v
    return d

  d = desc(0, 576 * 576, (576, 576, 1, 1, 1, 576, 576))            # 1x1 forward image, no padding: contiguous cast
  assert plan(ctypes.byref(d)) == (576 * 576 + per - 1) // per and d.a[7] == 1
  d = desc(1, 576 * 576, (576, 576, 1, 1, 1, 576, 576))            # 1x1 data-gradient image: tiled transpose, 64 x 32 tiles
  assert plan(ctypes.byref(d)) == (576 // 64) * (576 // 32) and d.a[7] == 2
  d = desc(0, 24 * 24 * 9 * 24, (576, 24, 3, 3, 24, 24, 24))       # grouped 3x3: element-wise gather
  assert plan(ctypes.byref(d)) == (24 * 24 * 9 * 24 + per - 1) // per and d.a[7] == 0
  d = desc(0, 8 * 32, (7, 30, 1, 1, 1, 32, 8))                     # padded 1x1: not a plain copy
  plan(ctypes.byref(d))
  assert d.a[7] == 0
  d = desc(2, 70 * 96, (70, 96, 1))                                # transposed pack2d
  assert plan(ctypes.byref(d)) == 2 * 3 and d.a[7] == 2


def test_workspace_bytes_is_the_size_of_the_unconstrained_plan(L):
  """tfpp_workspace_bytes: slices x output of the plan the dispatcher makes when the workspace does not limit it."""
  def ws(op, p, dt=BF16):
    out = ctypes.c_int64(-1)
    L.tfpp_workspace_bytes(op, ctypes.byref(p), dt, ctypes.byref(out))  # raises TfppError on a non-zero return code
    return out.value

  p = conv(12, 8, 8, 1512, 128, k=3, ws=False)  # K = 13608 over 6 output tiles: split-K
  p2 = conv(12, 8, 8, 1512, 128, k=3)
  p2.splitk_ws_floats = 1 << 40
  s = splits(L, p2)
  assert s > 1 and ws(0, p) == s * 12 * 8 * 8 * 128 * 4
  assert ws(0, conv(3840, 1, 1, 1512, 6048)) == 0  # 720 workgroups: no K split, no workspace
  w = wgrad(12, 16, 64, 576, 576, ws=False)
  w2 = wgrad(12, 16, 64, 576, 576)
  w2.ws_floats = 1 << 40
  plan = wplan(L, w2)
  assert plan[1] > 1 and plan[2] == 1 and ws(1, w) == plan[1] * 576 * 576 * 4
  assert ws(1, wgrad(3840, 1, 1, 1512, 6048)) == 0  # 576 tiles of 128x128: one slice, written straight into the gradient
  c = ctypes.c_int(576)
  assert ws(2, c) == L.raw('tfpp_bn_scratch_floats')(576) * 4 and ws(3, c) == L.raw('tfpp_reduce_scratch_floats')(1, 576) * 4
