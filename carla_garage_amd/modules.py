"""Parameter containers with the reference's ``state_dict`` schema.

These ``nn.Module`` trees only HOLD parameters / buffers under the reference's names (so released checkpoints,
``create_optimizer_groups``' isinstance checks, DDP, ``requires_grad_`` on sub-modules and ``SyncBatchNorm``
conversion keep working, team_code/train.py:479-531) and initialise them the way the reference does.  Their
``forward`` methods are never the compute path: all arithmetic goes through carla_garage_amd/engine.py -> HIP.
"""
import math

import numpy as np
import torch
from torch import nn

REGNETY_032 = dict(widths=(72, 216, 576, 1512), depths=(2, 5, 13, 1), group_w=24, stem_w=32, se_ratio=0.25)


def _no_forward(self, *a, **k):
  raise RuntimeError('parameter container: compute runs through carla_garage_amd.engine (HIP), not nn.Module.forward')


class FocalLossWeights(nn.Module):
  """Container with the state_dict schema of team_code/focal_loss.py::FocalLoss (one buffer: ``nll_loss.weight``)."""
  forward = _no_forward

  def __init__(self, alpha, gamma):
    super().__init__()
    self.gamma = gamma
    self.nll_loss = nn.NLLLoss(weight=alpha, reduction='none')


class ConvBn(nn.Module):
  """timm ConvNormAct naming: ``conv`` (no bias) + ``bn`` (team_code/model.py:586-589 relies on 'conv.' / '.bn')."""
  forward = _no_forward

  def __init__(self, cin, cout, k, stride=1, groups=1):
    super().__init__()
    self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False)
    self.bn = nn.BatchNorm2d(cout)
    fan_out = k * k * cout // groups
    nn.init.normal_(self.conv.weight, 0.0, math.sqrt(2.0 / fan_out))


class SqueezeExcite(nn.Module):
  forward = _no_forward

  def __init__(self, chs, rd):
    super().__init__()
    self.fc1 = nn.Conv2d(chs, rd, 1)
    self.fc2 = nn.Conv2d(rd, chs, 1)


class Bottleneck(nn.Module):
  """RegNet-Y block: conv1 1x1, conv2 grouped 3x3 (stride), se, conv3 1x1, optional downsample 1x1 (stride)."""
  forward = _no_forward

  def __init__(self, cin, cout, stride, group_w, se_ratio):
    super().__init__()
    self.conv1 = ConvBn(cin, cout, 1)
    self.conv2 = ConvBn(cout, cout, 3, stride, cout // group_w)
    self.se = SqueezeExcite(cout, int(round(cin * se_ratio)))
    self.conv3 = ConvBn(cout, cout, 1)
    if cin != cout or stride != 1:
      self.downsample = ConvBn(cin, cout, 1, stride)
    else:
      self.downsample = None
    nn.init.zeros_(self.conv3.bn.weight)  # zero_init_last
    self.stride = stride


class FeatureInfo:

  def __init__(self, info):
    self.info = info


class RegNetY(nn.ModuleDict):
  """``timm.create_model('regnety_032', features_only=True)``-shaped container: children stem, s1..s4."""
  forward = _no_forward

  def __init__(self, in_chans=3, arch=None):
    super().__init__()
    a = arch or REGNETY_032
    self['stem'] = ConvBn(in_chans, a['stem_w'], 3, 2)
    cin = a['stem_w']
    for i, (w, d) in enumerate(zip(a['widths'], a['depths'])):
      stage = nn.Sequential()
      for k in range(d):
        stage.add_module(f'b{k + 1}', Bottleneck(cin, w, 2 if k == 0 else 1, a['group_w'], a['se_ratio']))
        cin = w
      self[f's{i + 1}'] = stage
    self.in_chans = in_chans
    self.return_layers = {n: str(i) for i, n in enumerate(['stem', 's1', 's2', 's3', 's4'])}
    self.feature_info = FeatureInfo([dict(num_chs=a['stem_w'], reduction=2, module='stem')] + [
        dict(num_chs=w, reduction=4 * 2**i, module=f's{i + 1}') for i, w in enumerate(a['widths'])
    ])


class SelfAttention(nn.Module):
  forward = _no_forward

  def __init__(self, c):
    super().__init__()
    self.key = nn.Linear(c, c)
    self.query = nn.Linear(c, c)
    self.value = nn.Linear(c, c)
    self.proj = nn.Linear(c, c)


class AttentionWithWeights(SelfAttention):
  """Container of team_code/transfuser.py:404-443 (MultiheadAttentionWithAttention): key / query / value / proj linears; the two dropout modules carry
  the rates the engine reads (no parameters)."""

  def __init__(self, c, pdrop):
    super().__init__(c)
    self.attn_drop = nn.Dropout(pdrop)
    self.resid_drop = nn.Dropout(pdrop)


class DecoderLayerWithAttention(nn.Module):
  """Container of team_code/transfuser.py:447-477 (TransformerDecoderLayerWithAttention; the decoder of config.tp_attention)."""
  forward = _no_forward

  def __init__(self, d_model, dim_feedforward=2048, dropout=0.1, layer_norm_eps=1e-5):
    super().__init__()
    self.self_attn = AttentionWithWeights(d_model, dropout)
    self.multihead_attn = AttentionWithWeights(d_model, dropout)
    self.linear1 = nn.Linear(d_model, dim_feedforward)
    self.dropout = nn.Dropout(dropout)
    self.linear2 = nn.Linear(dim_feedforward, d_model)
    self.norm1 = nn.LayerNorm(d_model, eps=layer_norm_eps)
    self.norm2 = nn.LayerNorm(d_model, eps=layer_norm_eps)
    self.norm3 = nn.LayerNorm(d_model, eps=layer_norm_eps)
    self.dropout1 = nn.Dropout(dropout)
    self.dropout2 = nn.Dropout(dropout)
    self.dropout3 = nn.Dropout(dropout)
    self.activation = nn.GELU()  # a module: it survives the per-layer deep copy (transfuser.py:485), so this decoder really runs the exact GELU


class DecoderWithAttention(nn.Module):
  """Container of team_code/transfuser.py:479-508 (TransformerDecoderWithAttention): layers.{l}, norm."""
  forward = _no_forward

  def __init__(self, d_model, num_layers, norm):
    super().__init__()
    self.layers = nn.ModuleList([DecoderLayerWithAttention(d_model) for _ in range(num_layers)])
    self.num_layers = num_layers
    self.norm = norm


class Block(nn.Module):
  forward = _no_forward

  def __init__(self, c, block_exp):
    super().__init__()
    self.ln1 = nn.LayerNorm(c)
    self.ln2 = nn.LayerNorm(c)
    self.attn = SelfAttention(c)
    self.mlp = nn.Sequential(nn.Linear(c, block_exp * c), nn.ReLU(True), nn.Linear(block_exp * c, c), nn.Dropout(0.0))


class GPT(nn.Module):
  """Fusion transformer container (team_code/transfuser.py:260-299): pos_emb, blocks.{l}, ln_f."""
  forward = _no_forward

  def __init__(self, c, config, n_tokens):
    super().__init__()
    self.n_embd = c
    self.pos_emb = nn.Parameter(torch.zeros(1, n_tokens, c))
    self.blocks = nn.Sequential(*[Block(c, config.block_exp) for _ in range(config.n_layer)])
    self.ln_f = nn.LayerNorm(c)
    std, mean = getattr(config, 'gpt_linear_layer_init_std', 0.02), getattr(config, 'gpt_linear_layer_init_mean', 0.0)
    for m in self.modules():
      if isinstance(m, nn.Linear):
        m.weight.data.normal_(mean=mean, std=std)
        m.bias.data.zero_()
      elif isinstance(m, nn.LayerNorm):
        m.bias.data.zero_()
        m.weight.data.fill_(getattr(config, 'gpt_layer_norm_init_weight', 1.0))


SWIN3D_TINY = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=(8, 7, 7), patch_size=(2, 4, 4), mlp_ratio=4)


class WindowAttention3D(nn.Module):
  """Container for team_code/video_swin_transformer.py:87-137: relative_position_bias_table (parameter), relative_position_index
  (buffer, part of the state_dict), qkv, proj."""
  forward = _no_forward

  def __init__(self, dim, window_size, num_heads):
    super().__init__()
    self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
    wd, wh, ww = window_size
    self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing='ij')).flatten(1)  # 3, n
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= 2 * ww - 1
    self.register_buffer('relative_position_index', rel.sum(-1))
    self.qkv = nn.Linear(dim, dim * 3, bias=True)
    self.proj = nn.Linear(dim, dim)
    nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)


class SwinMlp(nn.Module):
  forward = _no_forward

  def __init__(self, dim, hidden):
    super().__init__()
    self.fc1 = nn.Linear(dim, hidden)
    self.fc2 = nn.Linear(hidden, dim)


class SwinBlock3D(nn.Module):
  forward = _no_forward

  def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio):
    super().__init__()
    self.window_size, self.shift_size = window_size, shift_size
    self.norm1 = nn.LayerNorm(dim)
    self.attn = WindowAttention3D(dim, window_size, num_heads)
    self.norm2 = nn.LayerNorm(dim)
    self.mlp = SwinMlp(dim, int(dim * mlp_ratio))


class PatchMerging(nn.Module):
  forward = _no_forward

  def __init__(self, dim):
    super().__init__()
    self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
    self.norm = nn.LayerNorm(4 * dim)


class SwinLayer(nn.Module):
  forward = _no_forward

  def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, downsample):
    super().__init__()
    shift = tuple(i // 2 for i in window_size)
    self.blocks = nn.ModuleList(
        [SwinBlock3D(dim, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else shift, mlp_ratio) for i in range(depth)])
    self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed3D(nn.Module):
  forward = _no_forward

  def __init__(self, patch_size, in_chans, embed_dim):
    super().__init__()
    self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
    self.norm = nn.LayerNorm(embed_dim)  # patch_norm=True is the reference's default (video_swin_transformer.py:510)


class CustomNorm(nn.Module):
  forward = _no_forward

  def __init__(self, num_features):
    super().__init__()
    self.norm = nn.LayerNorm(num_features)


class SwinTransformer3D(nn.Module):
  """Container for team_code/video_swin_transformer.py:470-585 as transfuser.py:44-50 builds it (Swin-T, in_chans = 1): children
  patch_embed, layers.layer0..3, norm.  The feature taps transfuser.py iterates (return_layers, :553-560) are layer0..2 (after their
  PatchMerging) and layer3 + norm: 192 / 384 / 768 / 768 channels at 1/8, 1/16, 1/32, 1/32 resolution, 3 time frames."""
  forward = _no_forward

  def __init__(self, in_chans=1, arch=None):
    super().__init__()
    a = arch or SWIN3D_TINY
    self.arch = a
    self.patch_embed = PatchEmbed3D(a['patch_size'], in_chans, a['embed_dim'])
    self.layers = nn.ModuleDict()
    n = len(a['depths'])
    for i in range(n):
      self.layers[f'layer{i}'] = SwinLayer(a['embed_dim'] * 2**i, a['depths'][i], a['num_heads'][i], a['window_size'], a['mlp_ratio'], i < n - 1)
    self.num_features = a['embed_dim'] * 2**(n - 1)
    self.norm = CustomNorm(self.num_features)
    self.feature_chs = [a['embed_dim'] * 2**min(i + 1, n - 1) for i in range(n)]  # 192, 384, 768, 768
    # init_weights is never called by the reference (transfuser.py:45-47): nn defaults + the trunc_normal bias tables


class TransfuserBackbone(nn.Module):
  """Container for team_code/transfuser.py:16-129 (default 2-D RegNet LiDAR branch)."""
  forward = _no_forward

  def __init__(self, config):
    super().__init__()
    if config.image_architecture != 'regnety_032' or config.lidar_architecture not in ('regnety_032', 'video_swin_tiny'):
      raise ValueError('the MI355X path implements the regnety_032 image branch and the regnety_032 / video_swin_tiny LiDAR branches '
                       f'(got {config.image_architecture} / {config.lidar_architecture})')
    self.config = config
    self.lidar_video = config.lidar_architecture == 'video_swin_tiny'
    in_ch = config.lidar_seq_len * (2 if config.use_ground_plane else 1)
    self.image_encoder = RegNetY(3)
    widths = REGNETY_032['widths']
    if self.lidar_video:  # transfuser.py:44-50,66-81: 3 time frames per scale, Conv3d 1x1x1 channel adapters
      self.lidar_encoder = SwinTransformer3D(1 + int(config.use_ground_plane))
      lidar_chs, frames = self.lidar_encoder.feature_chs, 3
      conv = lambda a, b: nn.Conv3d(a, b, kernel_size=1)
    else:
      self.lidar_encoder = RegNetY(in_ch)
      lidar_chs, frames = widths, 1
      conv = lambda a, b: nn.Conv2d(a, b, 1)
    self.lidar_time_frames = frames
    n_tok = config.img_vert_anchors * config.img_horz_anchors + frames * config.lidar_vert_anchors * config.lidar_horz_anchors
    self.transformers = nn.ModuleList([GPT(c, config, n_tok) for c in widths])
    self.lidar_channel_to_img = nn.ModuleList([conv(lc, c) for lc, c in zip(lidar_chs, widths)])
    self.img_channel_to_lidar = nn.ModuleList([conv(c, lc) for lc, c in zip(lidar_chs, widths)])
    self.num_image_features = widths[-1]
    self.num_features = lidar_chs[-1]
    self.perspective_upsample_factor = 32 // config.perspective_downsample_factor
    ch = config.bev_features_chanels
    if config.detect_boxes or config.use_bev_semantic:
      self.up_conv5 = nn.Conv2d(ch, ch, 3, padding=1)
      self.up_conv4 = nn.Conv2d(ch, ch, 3, padding=1)
      self.c5_conv = nn.Conv2d(lidar_chs[-1], ch, 1)


class UpsamplingConcat(nn.Module):
  """Container for team_code/bev_encoder.py:252-272: two 3x3 convs (no bias), each followed by InstanceNorm2d + ReLU (no parameters)."""
  forward = _no_forward

  def __init__(self, cin, cout):
    super().__init__()
    self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.InstanceNorm2d(cout), nn.ReLU(inplace=True),
                              nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.InstanceNorm2d(cout), nn.ReLU(inplace=True))


class BevEncoder(nn.Module):
  """Container for team_code/bev_encoder.py:15-137 (SimpleBEV-style lift of the image features into the BEV grid, LiDAR concatenated in BEV,
  a second RegNet on the fused grid).  Both RegNets lose their last stage (:36-38,77-79)."""
  forward = _no_forward

  def __init__(self, config):
    super().__init__()
    if config.image_architecture != 'regnety_032' or config.lidar_architecture != 'regnety_032':
      raise ValueError('the MI355X path implements bev_encoder with regnety_032 image / BEV networks '
                       f'(got {config.image_architecture} / {config.lidar_architecture})')
    self.config = config
    grid, valid = projection_grid(config)
    self.grid = nn.Parameter(grid, requires_grad=False)
    self.bev_projection_normalizer = nn.Parameter(torch.finfo(torch.float32).eps + valid.sum(dim=3).unsqueeze(1), requires_grad=False)
    self.valid_bev_pixels = nn.Parameter(valid.max(dim=3)[0].unsqueeze(1).transpose(2, 3).contiguous(), requires_grad=False)
    in_ch = config.lidar_seq_len * (2 if config.use_ground_plane else 1)
    self.image_encoder = RegNetY(3)
    del self.image_encoder['s4']
    self.bev_encoder = RegNetY(in_ch + config.bev_latent_dim)
    del self.bev_encoder['s4']
    widths = REGNETY_032['widths']
    self.num_features = widths[2]
    self.perspective_upsample_factor = 16 // config.perspective_downsample_factor
    ch = config.bev_features_chanels
    if config.detect_boxes or config.use_bev_semantic:
      self.up_conv5 = nn.Conv2d(ch, ch, 3, padding=1)
      self.up_conv4 = nn.Conv2d(ch, ch, 3, padding=1)
      self.c5_conv = nn.Conv2d(self.num_features, ch, 1)
    self.upsampling_layer = UpsamplingConcat(widths[1] + widths[2], config.image_u_net_output_features)
    self.depth_layer = nn.Conv2d(config.image_u_net_output_features, config.bev_latent_dim, kernel_size=1)
    self.bev_compressor = nn.Sequential(nn.Conv2d(config.bev_latent_dim, config.bev_latent_dim, 3, padding=1, bias=False),
                                        nn.InstanceNorm2d(config.bev_latent_dim), nn.GELU())
    self.num_image_features = config.bev_latent_dim


class AIMBackbone(nn.Module):
  """Container for team_code/aim.py:10-30: one RegNetY-3.2GF image encoder, no LiDAR branch, no fusion (BASELINE config 1)."""
  forward = _no_forward

  def __init__(self, config):
    super().__init__()
    if config.image_architecture != 'regnety_032':
      raise ValueError(f'the MI355X path implements the regnety_032 image branch (got {config.image_architecture})')
    self.config = config
    self.image_encoder = RegNetY(3)
    self.num_image_features = REGNETY_032['widths'][-1]
    self.num_features = REGNETY_032['widths'][-1]


class LidarCenterNetHead(nn.Module):
  """Container for team_code/center_net.py:23-47: 5 branches, plus velocity / brake when the input is temporal (:29-31)."""
  forward = _no_forward
  BRANCHES = ('heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res')

  def __init__(self, config):
    super().__init__()
    self.config = config
    c = config.bb_input_channel
    outs = dict(heatmap=config.num_bb_classes, wh=2, offset=2, yaw_class=config.num_dir_bins, yaw_res=1)
    if not (config.lidar_seq_len == 1 and config.seq_len == 1):
      self.BRANCHES = self.BRANCHES + ('velocity', 'brake')
      outs.update(velocity=1, brake=2)
    for n in self.BRANCHES:
      setattr(self, n + '_head', nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(c, outs[n], 1)))
    self.out_channels = outs

  def get_bboxes(self, center_heatmap_preds, wh_preds, offset_preds, yaw_class_preds, yaw_res_preds, velocity_preds=None, brake_preds=None):
    """team_code/center_net.py:142-170 -> decode_heatmap (172-237) as one HIP launch: (B, k, 9) boxes in image coordinates
    (x, y, w, h, yaw, velocity, brake, class, score), top-k by score.  Single-frame configuration (velocity = brake = 0)."""
    import torch
    from . import ops
    from ._lib import lib
    cfg = self.config
    heat = center_heatmap_preds
    if not heat.is_cuda:
      raise RuntimeError('carla_garage_amd runs on the MI355X HIP path only: CenterNet decode needs CUDA tensors')
    maps = [t.detach().float().contiguous() for t in (heat, wh_preds, offset_preds, yaw_class_preds, yaw_res_preds)]
    B, ncls, H, W = maps[0].shape
    k = int(getattr(cfg, 'top_k_center_keypoints', 100))
    if int(getattr(cfg, 'center_net_max_pooling_kernel', 3)) != 3:
      raise NotImplementedError('the decode kernel implements the reference default 3x3 local-maximum window')
    out = torch.empty((B, k, 9), device=heat.device, dtype=torch.float32)
    lib.load()
    lib.tfpp_centernet_decode(*[ops.ptr(m) for m in maps], ops.ptr(out), B, ncls, H, W, k, int(cfg.num_dir_bins),
                              float(cfg.lidar_resolution_width / W), float(cfg.lidar_resolution_height / H), ops.stream())
    return out


class PerspectiveDecoder(nn.Module):
  """Container for team_code/transfuser_utils.py:668-695."""
  forward = _no_forward

  def __init__(self, cin, cout, c0, c1, c2, scale_factor_0, scale_factor_1):
    super().__init__()
    self.scale_factor_0, self.scale_factor_1 = scale_factor_0, scale_factor_1
    self.deconv1 = nn.Sequential(nn.Conv2d(cin, c0, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c0, c1, 3, 1, 1), nn.ReLU(True))
    self.deconv2 = nn.Sequential(nn.Conv2d(c1, c2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c2, c2, 3, 1, 1), nn.ReLU(True))
    self.deconv3 = nn.Sequential(nn.Conv2d(c2, c2, 3, 1, 1), nn.ReLU(True), nn.Conv2d(c2, cout, 3, 1, 1))


class GRUWaypointsPredictorInterFuser(nn.Module):
  """Container for team_code/model.py:839-855."""
  forward = _no_forward

  def __init__(self, input_dim, waypoints, hidden_size, target_point_size):
    super().__init__()
    self.gru = nn.GRU(input_size=input_dim, hidden_size=hidden_size, batch_first=True)
    if target_point_size > 0:
      self.encoder = nn.Linear(target_point_size, hidden_size)
    self.target_point_size, self.hidden_size, self.waypoints = target_point_size, hidden_size, waypoints
    self.decoder = nn.Linear(hidden_size, 2)


def projection_grid(config):
  """team_code/transfuser_utils.py:596-665 (create_projection_grid): for every voxel centre of the (depth, width, height) grid around the car
  the camera pixel a pinhole projection puts it on, in the reference's normalised coordinates ((u, v) / (0.5 size - 0.5) - 1, third
  component 0), and the mask of voxels that land inside the image in front of the camera.  Returns grid (1, d, w, h, 3), valid (1, d, w, h)."""
  mpp = 1.0 / config.pixels_per_meter
  xs = torch.arange(config.min_x, config.max_x, mpp) + 0.5 * mpp  # lateral (width)
  ys = torch.arange(config.min_y, config.max_y, mpp) + 0.5 * mpp  # forward (depth)
  mz = mpp * config.bev_grid_height_downsample_factor
  zs = torch.arange(config.min_z_projection, config.max_z_projection, mz) + 0.5 * mz
  fwd, lat, up = torch.meshgrid(ys, xs, zs, indexing='ij')
  d, w, h = fwd.shape
  cam = torch.tensor(config.camera_pos, dtype=torch.float32)
  pts = torch.stack((fwd, lat, up), 0).reshape(3, -1) - cam.unsqueeze(1)
  f = config.camera_width / (2.0 * np.tan(config.camera_fov * np.pi / 360.0))
  intr = torch.from_numpy(np.array([[f, 0.0, config.camera_width / 2.0], [0.0, f, config.camera_height / 2.0],
                                    [0.0, 0.0, 1.0]])).to(torch.float32)
  proj = intr @ torch.stack((pts[1], pts[2], pts[0]))
  depth = proj[2:3]
  grid = torch.zeros_like(proj)
  grid[:2] = proj[:2] / depth
  grid = grid.view(3, d, w, h)
  inside = (grid[0:1] >= 0.0) & (grid[0:1] < config.camera_width) & (grid[1:2] >= 0.0) & (grid[1:2] < config.camera_height) & \
      (depth.view(1, d, w, h) > 0.0)
  grid[0:1] = grid[0:1] / (0.5 * config.camera_width - 0.5) - 1.0
  grid[1:2] = grid[1:2] / (0.5 * config.camera_height - 0.5) - 1.0
  grid = grid.reshape(1, 3, d, w, h, 1).transpose(1, 5).squeeze(1)
  return grid.contiguous(), inside.to(torch.float32)


def visibility_mask(config):
  """``valid_bev_pixels`` (1,1,H,W): BEV pixels with at least one voxel centre that a pinhole projection puts
  inside the camera image -- team_code/transfuser_utils.py:596-665 + team_code/model.py:93-98.  Init-time, host."""
  mpp = 1.0 / config.pixels_per_meter
  xs = torch.arange(config.min_x, config.max_x, mpp) + 0.5 * mpp  # lateral
  ys = torch.arange(config.min_y, config.max_y, mpp) + 0.5 * mpp  # forward (depth)
  mz = mpp * config.bev_grid_height_downsample_factor
  zs = torch.arange(config.min_z_projection, config.max_z_projection, mz) + 0.5 * mz
  fwd, lat, up = torch.meshgrid(ys, xs, zs, indexing='ij')
  cam = torch.tensor(config.camera_pos, dtype=torch.float32)
  pts = torch.stack((fwd, lat, up), 0).reshape(3, -1) - cam.unsqueeze(1)
  f = config.camera_width / (2.0 * np.tan(config.camera_fov * np.pi / 360.0))
  intr = torch.from_numpy(np.array([[f, 0.0, config.camera_width / 2.0], [0.0, f, config.camera_height / 2.0],
                                    [0.0, 0.0, 1.0]])).to(torch.float32)
  proj = intr @ torch.stack((pts[1], pts[2], pts[0]))
  depth = proj[2:3]
  uv = proj[:2] / depth
  inside = (uv[0:1] >= 0.0) & (uv[0:1] < config.camera_width) & (uv[1:2] >= 0.0) & (uv[1:2] < config.camera_height) & (depth > 0.0)
  vol = inside.to(torch.float32).reshape(1, len(ys), len(xs), len(zs))
  return vol.max(dim=3)[0].unsqueeze(1).transpose(2, 3).contiguous()


def sine_position_table(h, w, num_pos_feats, temperature=10000.0):
  """Constant of ``PositionEmbeddingSine(num_pos_feats, normalize=True)`` (team_code/model.py:916-953) for an
  h x w grid, returned token-major [h*w, 2*num_pos_feats] (y half then x half).  Init-time, host."""
  two_pi = 2.0 * math.pi
  yy = (torch.arange(1, h + 1, dtype=torch.float32) / (float(h) + 1e-6) * two_pi).view(h, 1, 1).expand(h, w, 1)
  xx = (torch.arange(1, w + 1, dtype=torch.float32) / (float(w) + 1e-6) * two_pi).view(1, w, 1).expand(h, w, 1)
  i = torch.arange(num_pos_feats, dtype=torch.float32)
  freq = temperature**(2 * torch.div(i, 2, rounding_mode='floor') / num_pos_feats)

  def enc(v):
    a = v / freq
    return torch.stack((a[..., 0::2].sin(), a[..., 1::2].cos()), -1).flatten(-2)

  return torch.cat((enc(yy), enc(xx)), -1).reshape(h * w, 2 * num_pos_feats).contiguous()
