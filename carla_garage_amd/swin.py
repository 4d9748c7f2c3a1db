"""Video-Swin (Swin-T, 3-D shifted windows) LiDAR branch on the HIP path -- BASELINE config 5.

Follows team_code/video_swin_transformer.py as team_code/transfuser.py:44-50,151-186 uses it: PatchEmbed3D (:427-467), BasicLayer
(:349-424) of SwinTransformerBlock3D (:173-288) with WindowAttention3D (:87-170), PatchMerging (:291-325), CustomNorm (:587-601).
Tokens live as rows [B * D * H * W, C] (the reference's (B, D, H, W, C) layout, channels contiguous).  Zero padding to window
multiples, the cyclic shift, window partition / reverse and the 2x2 merge are single gather launches over index tables built once per
geometry on the host (the reference caches its shift masks the same way, :329-342); linears are the implicit-GEMM kernels, the
147 x 147 window attention is two strided batched GEMMs around tfpp_softmax_window_bias.

Training: every step records its adjoint on the engine's tape.  The gathers are permutations (plus zero padding), so their adjoints are
the same kernel with the inverse table; the attention core keeps P for backward (dP = dO V^T, dV = P^T dO, dS through tfpp_softmax_bwd,
dQ = dS K, dK = dS^T Q, bias-table gradient by tfpp_window_bias_grad); DropPath (stochastic depth, rates 0 .. 0.2 over the 12 blocks,
:535) is tfpp_drop_path on the residual branch in window / token layout (one draw per sample)."""
import math
import os

import torch

from . import ops
from ._lib import ACT_GELU, ACT_NONE

FUSED_WINDOW_ATTN = os.environ.get('TFPP_FUSED_WINDOW_ATTN', '1') != '0'  # bf16: fused LDS window attention (0: bgemm -> softmax -> bgemm)
DROP_PATH_RATE = 0.2  # SwinTransformer3D's default, which transfuser.py:45-47 does not override


def window_geometry(dims, window_size, shift_size):
  """video_swin_transformer.py:72-84 (get_window_size): a window never exceeds the feature map; no shift along a clipped axis."""
  ws, ss = list(window_size), list(shift_size)
  for i in range(3):
    if dims[i] <= window_size[i]:
      ws[i], ss[i] = dims[i], 0
  return tuple(ws), tuple(ss)


def _partition(x, ws):  # (Dp, Hp, Wp) -> (nW, n)   (:40-53)
  d, h, w = x.shape
  x = x.view(d // ws[0], ws[0], h // ws[1], ws[1], w // ws[2], ws[2])
  return x.permute(0, 2, 4, 1, 3, 5).reshape(-1, ws[0] * ws[1] * ws[2])


def _reverse(win, ws, d, h, w):  # (nW, n) -> (Dp, Hp, Wp)   (:56-69)
  x = win.view(d // ws[0], h // ws[1], w // ws[2], ws[0], ws[1], ws[2])
  return x.permute(0, 3, 1, 4, 2, 5).reshape(d, h, w)


def window_maps(B, D, H, W, ws, ss, tail=8):
  """Index tables of one (geometry, shift) pair (host, int32):
    fwd [B * nW * n + tail]  source token row of every window slot, -1 for the zero padding (and for `tail` spare rows: the P.V product
                             reads K in whole 16-byte vectors);
    rev [B * D * H * W]      window slot holding the result of every real token (window_reverse + roll back + crop);
    mask [nW, n, n] fp32     0 / -100 shift mask (compute_mask, :329-342), None without shift."""
  Dp, Hp, Wp = (math.ceil(D / ws[0]) * ws[0], math.ceil(H / ws[1]) * ws[1], math.ceil(W / ws[2]) * ws[2])
  src = torch.full((Dp, Hp, Wp), -1, dtype=torch.int64)
  src[:D, :H, :W] = torch.arange(D * H * W).view(D, H, W)
  shifted = torch.roll(src, shifts=(-ss[0], -ss[1], -ss[2]), dims=(0, 1, 2)) if any(ss) else src
  fwd1 = _partition(shifted, ws)  # (nW, n)
  nW, n = fwd1.shape
  slots = _reverse(torch.arange(nW * n).view(nW, n), ws, Dp, Hp, Wp)
  if any(ss):
    slots = torch.roll(slots, shifts=ss, dims=(0, 1, 2))
  rev1 = slots[:D, :H, :W].reshape(-1)
  fwd = torch.cat([torch.where(fwd1 >= 0, fwd1 + b * D * H * W, fwd1).reshape(-1) for b in range(B)] + [torch.full((tail,), -1, dtype=torch.int64)])
  rev = torch.cat([rev1 + b * nW * n for b in range(B)])
  mask = None
  if any(ss):
    img = torch.zeros((Dp, Hp, Wp))
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
      for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
        for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
          img[d, h, w] = cnt
          cnt += 1
    mw = _partition(img, ws)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    mask = torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).contiguous()
  return fwd.int(), rev.int(), mask, nW, n


def merge_maps(B, D, H, W):
  """PatchMerging (:305-309): four index tables [B * D * H/2 * W/2] for x0 (even h, even w), x1 (odd h, even w), x2 (even h, odd w),
  x3 (odd h, odd w); -1 where an odd-sized map is zero-padded."""
  Hp, Wp = H + H % 2, W + W % 2
  src = torch.full((B, D, Hp, Wp), -1, dtype=torch.int64)
  src[:, :, :H, :W] = torch.arange(B * D * H * W).view(B, D, H, W)
  return [src[:, :, i::2, j::2].reshape(-1).int() for (i, j) in ((0, 0), (1, 0), (0, 1), (1, 1))], Hp // 2, Wp // 2


def merge_inverse_map(B, D, H, W):
  """Adjoint of merge_maps: for every input token the row of the concatenated [n_out * 4, C] view that holds its copy."""
  Hp, Wp = H + H % 2, W + W % 2
  out = torch.arange(B * D * (Hp // 2) * (Wp // 2)).view(B, D, Hp // 2, Wp // 2)
  inv = torch.empty((B, D, Hp, Wp), dtype=torch.int64)
  for j, (i0, j0) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
    inv[:, :, i0::2, j0::2] = out * 4 + j
  return inv[:, :, :H, :W].reshape(-1).int()


class VideoSwin:
  """Runs ``backbone.lidar_encoder`` (modules.SwinTransformer3D) through the engine's kernels."""

  def __init__(self, engine):
    self.e = engine
    self.enc = engine.m.backbone.lidar_encoder
    self._maps = {}
    self.drop_path_rate = DROP_PATH_RATE  # tests set 0 to compare with a reference whose DropPath modules are switched off
    self.taps = None  # tests: dict that receives the (B, D, H, W, C) output of the patch embedding and of every layer

  def build_specs(self):
    e, enc, p = self.e, self.enc, 'backbone.lidar_encoder'
    e._spec(f'{p}.patch_embed.proj', enc.patch_embed.proj.weight, enc.patch_embed.proj.bias)  # (96, 1, 2, 4, 4) -> K = 32 in (kt, kh, kw) order
    for lname, layer in enc.layers.items():
      for j, blk in enumerate(layer.blocks):
        q = f'{p}.layers.{lname}.blocks.{j}'
        e._spec(q + '.attn.qkv', blk.attn.qkv.weight, blk.attn.qkv.bias)
        e._spec(q + '.attn.proj', blk.attn.proj.weight, blk.attn.proj.bias)
        e._spec(q + '.mlp.fc1', blk.mlp.fc1.weight, blk.mlp.fc1.bias)
        e._spec(q + '.mlp.fc2', blk.mlp.fc2.weight, blk.mlp.fc2.bias)
      if layer.downsample is not None:
        e._spec(f'{p}.layers.{lname}.downsample.reduction', layer.downsample.reduction.weight)

  def _dev(self, key, fn):
    k = (key, str(self.e.device))
    if k not in self._maps:
      v = fn()
      self._maps[k] = tuple(t.to(self.e.device) if torch.is_tensor(t) else t for t in v) if isinstance(v, tuple) else v.to(self.e.device)
    return self._maps[k]

  def stem(self, lidar):
    """lidar: fp32 (B, T, H, W) -> tokens [B, T/2, H/4, W/4, 96] (patch_embed + norm; pos_drop is the identity at p = 0)."""
    e = self.e
    B, T, H, W = lidar.shape
    rows = ops.patchify3d(lidar, e.dtype)
    x = e.linear(rows, 'backbone.lidar_encoder.patch_embed.proj', x_grad=False)
    x = e.layernorm(x, self.enc.patch_embed.norm).view(B, T // 2, H // 4, W // 4, -1)
    if self.taps is not None:
      self.taps['swin_patch_embed'] = x
    return x

  def drop_path(self, y, samples, p):
    """timm DropPath on a residual branch (identity in eval mode / at rate 0)."""
    e = self.e
    if not e.training or p <= 0.0:
      return y
    seed = e.next_seed()
    out = ops.drop_path(y, samples, p, seed)
    if e.tape is not None:
      e.rec([out], [y], lambda d: ops.drop_path(d, samples, p, seed))
    return out

  def block(self, x, key, blk, B, D, H, W, dp_rate=0.0):
    """SwinTransformerBlock3D.forward (:262-288) on rows x [B*D*H*W, C]."""
    e = self.e
    C = x.shape[-1]
    ntok = B * D * H * W
    ws, ss = window_geometry((D, H, W), blk.window_size, blk.shift_size)
    fwd, rev, mask, nW, n = self._dev(('win', B, D, H, W, ws, ss), lambda: window_maps(B, D, H, W, ws, ss))
    heads = blk.attn.num_heads
    d = C // heads
    scale = d**-0.5
    rel = self._dev(('rel', id(blk.attn), n), lambda: blk.attn.relative_position_index[:n, :n].contiguous().int().cpu())  # (:151-152)
    table = blk.attn.relative_position_bias_table
    h = e.layernorm(x, blk.norm1)
    nwin = B * nW
    nslot = nwin * n
    xw = ops.gather_rows(h, fwd, nslot + 8, C)                         # pad + roll + window_partition (+ 8 spare zero rows)
    if e.tape is not None:
      e.rec([xw], [h], lambda dxw: ops.gather_rows(dxw, rev, ntok, C))  # adjoint: every real token reads the gradient of its slot
    qkv = e.linear(xw, key + '.attn.qkv')                              # [nslot + 8, 3C]: q | k | v, head-major inside each
    flat = qkv.view(-1)
    q, k, v = flat[0:], flat[C:], flat[2 * C:]
    npad = ops.pad_to(n, 8)
    sbs = (heads * n * npad, n * npad)
    O = torch.empty((nslot, C), device=x.device, dtype=x.dtype)
    if x.dtype == torch.bfloat16 and FUSED_WINDOW_ATTN:
      # one workgroup per (window, head, 64 queries): K, V and the 147 x 147 scores of a window stay on the CU (attention_kernels.hip);
      # training keeps the probabilities (bf16, pitch npad, pad columns zero) for the unfused backward below
      ld_b = ops.pad_to(n, 16)
      bias = ops.window_bias_dense(table.detach(), rel, heads, n, ld_b)
      maskp = None
      if mask is not None:
        maskp = self._dev(('maskp', B, D, H, W, ws, ss, ld_b), lambda: torch.nn.functional.pad(mask.cpu(), (0, ld_b - n)).contiguous())
      S = ops.zeros((nwin, heads, n, npad), x.dtype, x.device) if e.tape is not None else None
      ops.attn_window_fwd(q, k, v, O, bias, maskp, S, B=nwin, nh=heads, T=n, d=d, ld_q=3 * C, ld_kv=3 * C, ld_o=C, scale=scale)
    else:
      S = ops.zeros((nwin, heads, n, npad), x.dtype, x.device)         # pad columns stay 0: the P.V product runs K = npad
      ops.bgemm(q, k, S, M=n, N=n, K=d, lda=3 * C, ldb=3 * C, ldc=npad, batch0=nwin, batch1=heads, a_bs=(n * 3 * C, d), b_bs=(n * 3 * C, d), c_bs=sbs)
      ops.softmax_window_bias(S, table.detach(), rel, mask, nwin, heads, n, scale, ld=npad)
      ops.bgemm(S, v, O, M=n, N=d, K=npad, lda=npad, ldb=3 * C, ldc=C, batch0=nwin, batch1=heads, a_bs=sbs, b_bs=(n * 3 * C, d), c_bs=(n * C, d),
                b_km=True)
    if e.tape is not None:

      def bwd_attn(dO, P=S):
        dqkv = ops.zeros(tuple(qkv.shape), qkv.dtype, qkv.device)
        df = dqkv.view(-1)
        dq, dk, dv = df[0:], df[C:], df[2 * C:]
        kw = dict(batch0=nwin, batch1=heads)
        dP = ops.zeros((nwin, heads, n, npad), P.dtype, P.device)
        ops.bgemm(dO, v, dP, M=n, N=n, K=d, lda=C, ldb=3 * C, ldc=npad, a_bs=(n * C, d), b_bs=(n * 3 * C, d), c_bs=sbs, **kw)            # dP = dO V^T
        ops.bgemm(P, dO, dv, M=n, N=d, K=n, lda=npad, ldb=C, ldc=3 * C, a_bs=sbs, b_bs=(n * C, d), c_bs=(n * 3 * C, d), a_km=True, b_km=True,
                  **kw)                                                                                                                     # dV = P^T dO
        ops.softmax_bwd(P, dP, nwin * heads * n, n, npad, alpha=scale)                                                                      # dP := dS
        if table.requires_grad:
          ops.window_bias_grad(dP, rel, e.g(table), nwin, heads, n, npad, 1.0 / scale)
        ops.bgemm(dP, k, dq, M=n, N=d, K=npad, lda=npad, ldb=3 * C, ldc=3 * C, a_bs=sbs, b_bs=(n * 3 * C, d), c_bs=(n * 3 * C, d), b_km=True, **kw)  # dQ = dS K
        ops.bgemm(dP, q, dk, M=n, N=d, K=n, lda=npad, ldb=3 * C, ldc=3 * C, a_bs=sbs, b_bs=(n * 3 * C, d), c_bs=(n * 3 * C, d), a_km=True, b_km=True,
                  **kw)                                                                                                                     # dK = dS^T Q
        return dqkv

      e.rec([O], [qkv], bwd_attn)
    y = e.linear(O, key + '.attn.proj')
    y = self.drop_path(y, B, dp_rate)                                  # window layout: the windows of a sample are contiguous
    x2 = ops.gather_rows(y, rev, ntok, C, add=x)                       # window_reverse + roll back + crop + shortcut
    if e.tape is not None:
      fwd_slots = fwd[:nslot]
      e.rec([x2], [y, x], lambda dx2: (ops.gather_rows(dx2, fwd_slots, nslot, C), dx2))
    h = e.layernorm(x2, blk.norm2)
    if e.tape is None and not (e.training and dp_rate > 0.0):
      h = e.linear(h, key + '.mlp.fc1', act=ACT_GELU)
      return e.linear(h, key + '.mlp.fc2', act=ACT_NONE, res=x2)
    h1 = e.linear(h, key + '.mlp.fc1')
    a = ops.affine_act(h1, act=ACT_GELU)
    if e.tape is not None:
      e.rec([a], [h1], lambda da: ops.act_bwd(da, h1, ACT_GELU))      # exact GELU derivative needs the pre-activation
    m = e.linear(a, key + '.mlp.fc2')
    m = self.drop_path(m, B, dp_rate)
    return e.add(x2, m)

  def layer(self, i, x):
    """BasicLayer i (:404-424) (+ the final CustomNorm after layer 3, transfuser.py iterates 'layer3' and 'norm' as one block).
    x: [B, D, H, W, C] -> [B, D, H', W', C']."""
    e = self.e
    lname = f'layer{i}'
    layer = self.enc.layers[lname]
    B, D, H, W, C = x.shape
    rows = x.reshape(B * D * H * W, C)
    depths = self.enc.arch['depths']
    total = sum(depths)
    for j, blk in enumerate(layer.blocks):
      k = sum(depths[:i]) + j  # stochastic depth decay rule: linspace(0, rate, sum(depths)) (:535)
      rows = self.block(rows, f'backbone.lidar_encoder.layers.{lname}.blocks.{j}', blk, B, D, H, W, self.drop_path_rate * k / (total - 1))
    if layer.downsample is not None:
      maps = self._dev(('merge', B, D, H, W), lambda: tuple(merge_maps(B, D, H, W)[0]))
      H2, W2 = (H + H % 2) // 2, (W + W % 2) // 2
      n_out = B * D * H2 * W2
      cat = torch.empty((n_out, 4 * C), device=x.device, dtype=x.dtype)
      for j, idx in enumerate(maps):
        ops.gather_rows(rows, idx, n_out, C, out=cat, dst_ld=4 * C, dst_off=j * C)
      if e.tape is not None:
        inv = self._dev(('merge_inv', B, D, H, W), lambda: merge_inverse_map(B, D, H, W))
        ntok_in = B * D * H * W
        e.rec([cat], [rows], lambda dcat, C=C, inv=inv, ntok_in=ntok_in: ops.gather_rows(dcat.view(-1, C), inv, ntok_in, C))  # (C changes below)
      cat = e.layernorm(cat, layer.downsample.norm)
      rows = e.linear(cat, f'backbone.lidar_encoder.layers.{lname}.downsample.reduction')
      H, W, C = H2, W2, 2 * C
    if self.taps is not None:
      self.taps[f'swin_layer{i}'] = rows.view(B, D, H, W, C)
    if i == len(self.enc.layers) - 1:
      rows = e.layernorm(rows, self.enc.norm.norm)
    return rows.view(B, D, H, W, C)
