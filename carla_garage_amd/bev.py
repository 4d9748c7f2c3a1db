"""The ``bev_encoder`` backbone (team_code/bev_encoder.py) on the HIP path: image RegNet (3 stages) -> UpsamplingConcat U-Net step ->
32-channel depth layer -> camera-to-BEV lift (tfpp_bev_lift_*) -> bev_compressor -> concatenation with the LiDAR BEV -> a second RegNet
(3 stages) on the fused grid -> BEV feature pyramid.  InstanceNorm2d (no parameters) is BatchNorm arithmetic per sample: the existing
two-stage statistics / finalize / apply kernels are run on each sample's rows (12 x 3 small launches per norm at bs = 12 -- a dedicated
kernel is the obvious refinement)."""
import torch

from . import ops
from ._lib import ACT_GELU, ACT_NONE, ACT_RELU

F32 = torch.float32


class BevEncoderRunner:

  def __init__(self, engine):
    self.e = engine
    self.bb = engine.m.backbone
    self.taps = None  # tests: receives the depth-layer output, the lifted and the compressed BEV features

  def build_specs(self):
    e, bb, cfg = self.e, self.bb, self.e.cfg
    for br, enc, cin_store in (('image_encoder', bb.image_encoder, 8), ('bev_encoder', bb.bev_encoder, ops.pad_to(bb.bev_encoder.in_chans, 8))):
      p = f'backbone.{br}'
      e._spec(f'{p}.stem', enc['stem'].conv.weight, bn=enc['stem'].bn, stride=2, pad=1, cin_store=cin_store)
      for si in range(1, 4):
        for bname, blk in enc[f's{si}'].named_children():
          q = f'{p}.s{si}.{bname}'
          e._spec(q + '.conv1', blk.conv1.conv.weight, bn=blk.conv1.bn)
          e._spec(q + '.conv2', blk.conv2.conv.weight, bn=blk.conv2.bn, stride=blk.stride, pad=1, groups=blk.conv2.conv.groups)
          e._spec(q + '.conv3', blk.conv3.conv.weight, bn=blk.conv3.bn)
          if blk.downsample is not None:
            e._spec(q + '.downsample', blk.downsample.conv.weight, bn=blk.downsample.bn, stride=blk.stride)
    e._spec('backbone.upsampling_layer.conv.0', bb.upsampling_layer.conv[0].weight, pad=1)
    e._spec('backbone.upsampling_layer.conv.3', bb.upsampling_layer.conv[3].weight, pad=1)
    e._spec('backbone.depth_layer', bb.depth_layer.weight, bb.depth_layer.bias)
    e._spec('backbone.bev_compressor.0', bb.bev_compressor[0].weight, pad=1)

  # ------------------------------------------------------------------------------------------------ InstanceNorm2d (+ ReLU)
  def instance_norm(self, x, act=ACT_NONE, eps=1e-5):
    """nn.InstanceNorm2d(C) (affine=False, no running statistics) on NHWC x, optionally fused with ReLU (bev_encoder.py:120,262-267)."""
    e = self.e
    y, mean, invstd = ops.instance_norm_fwd(x, act, eps)
    if e.tape is not None:
      e.rec([y], [x], lambda dy: ops.instance_norm_bwd(dy, y, x, mean, invstd, act == ACT_RELU))
    return y

  def concat_channels(self, parts, c_store):
    """torch.cat(parts, dim=1) in NHWC with the result zero-padded to c_store channels; parts: [(tensor [B,H,W,ld], real channels)]."""
    e = self.e
    B, H, W, _ = parts[0][0].shape
    rows = B * H * W
    first = parts[0][0]
    if sum(c for _, c in parts) != c_store:
      out = ops.zeros((B, H, W, c_store), first.dtype, first.device)
    else:
      out = torch.empty((B, H, W, c_store), device=first.device, dtype=first.dtype)
    off = 0
    offs = []
    for t, c in parts:
      ops.copy_rows(t, out, rows, c, t.shape[-1], 0, c_store, off)
      offs.append(off)
      off += c
    if e.tape is not None:

      def bwd(d):
        grads = []
        for (t, c), o in zip(parts, offs):
          if t.dtype != d.dtype or not getattr(t, '_tfpp_wants_grad', True):
            grads.append(None)
            continue
          g = ops.zeros(tuple(t.shape), t.dtype, t.device) if t.shape[-1] != c else torch.empty_like(t)
          ops.copy_rows(d, g, rows, c, c_store, o, t.shape[-1], 0)
          grads.append(g)
        return tuple(grads)

      e.rec([out], [t for t, _ in parts], bwd)
    return out

  def lift(self, img):
    """bev_encoder.py:180-201: (B, Hf, Wf, 32) image features -> (B, 256, 256, 32) BEV features (transposed + masked)."""
    e, bb = self.e, self.bb
    B, Hf, Wf, C = img.shape
    _, D, Wd, Z, _ = bb.grid.shape

    def host_coords():
      g = bb.grid.detach().float().cpu()[0]  # (D, W, Z, 3): x = width pixel, y = height pixel, normalised (transfuser_utils.py:658-660)
      ix = ((g[..., 0] + 1.0) * Wf - 1.0) * 0.5  # grid_sample, align_corners=False
      iy = ((g[..., 1] + 1.0) * Hf - 1.0) * 0.5
      return torch.stack((ix, iy), -1).contiguous()

    def host_scale():
      norm = bb.bev_projection_normalizer.detach().float().cpu()[0, 0]  # (D, W)
      valid = bb.valid_bev_pixels.detach().float().cpu()[0, 0]          # (W, D): already transposed
      return (valid / norm.t()).contiguous()

    coords = e._const(f'bev_coords{Hf}x{Wf}', host_coords)
    scale = e._const('bev_scale', host_scale)
    out = ops.bev_lift_fwd(img, coords, scale, D, Wd, Z)
    if e.tape is not None:
      e.rec([out], [img], lambda d: ops.cast(ops.bev_lift_bwd(d.view(out.shape), coords, scale, Hf, Wf, D, Wd, Z), img.dtype))
    return out

  def forward(self, xi, lidar_in):
    """xi: normalised NHWC image (8 storage channels); lidar_in: fp32 NCHW LiDAR BEV.  Returns (image feature grid for the perspective
    decoders, fused BEV features (B, 16, 16, 576))."""
    e, bb, cfg = self.e, self.bb, self.e.cfg
    dt_ = e.dtype
    ACT_RELU_ = ACT_RELU
    xi = e.conv(xi, 'backbone.image_encoder.stem', act=ACT_RELU_, x_grad=False)
    xi = e.stage(xi, 'backbone.image_encoder.s1', bb.image_encoder['s1'])
    f2 = e.stage(xi, 'backbone.image_encoder.s2', bb.image_encoder['s2'])
    f3 = e.stage(f2, 'backbone.image_encoder.s3', bb.image_encoder['s3'])
    up = e.upsample(f3, f2.shape[1], f2.shape[2])                       # UpsamplingConcat (:269-272): cat([x, upsampled], dim=1)
    cat = self.concat_channels([(f2, f2.shape[-1]), (up, up.shape[-1])], f2.shape[-1] + up.shape[-1])
    y = e.conv(cat, 'backbone.upsampling_layer.conv.0')
    y = self.instance_norm(y, ACT_RELU_)
    y = e.conv(y, 'backbone.upsampling_layer.conv.3')
    y = self.instance_norm(y, ACT_RELU_)
    img = e.conv(y, 'backbone.depth_layer')
    bev = self.lift(img)
    if self.taps is not None:
      self.taps['depth'], self.taps['lifted'] = img, bev
    bev = e.conv(bev, 'backbone.bev_compressor.0')
    bev = self.instance_norm(bev)
    pre = bev
    bev = ops.affine_act(pre, act=ACT_GELU)
    if e.tape is not None:
      e.rec([bev], [pre], lambda d: ops.act_bwd(d, pre, ACT_GELU))
    if self.taps is not None:
      self.taps['compressed'] = bev
    c_l = lidar_in.shape[1]
    lid = ops.nchw_to_nhwc_affine(lidar_in, dt_, ops.pad_to(c_l, 8))     # (B, 256, 256, 8): LiDAR channels first 1..., rest zero
    lid._tfpp_wants_grad = False
    fused = self.concat_channels([(bev, bev.shape[-1]), (lid, c_l)], ops.pad_to(bev.shape[-1] + c_l, 8))
    xl = e.conv(fused, 'backbone.bev_encoder.stem', act=ACT_RELU_)
    for si in (1, 2, 3):
      xl = e.stage(xl, f'backbone.bev_encoder.s{si}', bb.bev_encoder[f's{si}'])
    return img, xl
