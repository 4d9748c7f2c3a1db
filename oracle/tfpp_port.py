"""CPU oracle ("port"): functional plain-PyTorch fp32 restatement of the reference's
TransFuser++ hot path -- ``LidarCenterNet.forward`` and ``compute_loss`` for the
default ``GlobalConfig()`` (backbone 'transFuser', regnety_032 x2,
transformer_decoder_join, controller-input prediction; optional WP-GRU variant).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the checker
/ the timed CPU baseline -- never by the product.

It operates on a flat ``state_dict`` with the reference's key schema
 instead of a module tree, travels to the GPU box (where
``/root/reference`` does not exist), and is pinned against the reference itself:
tests/test_oracle.py compares it with the unmodified reference imported from
``/root/reference`` (build container) and with tests/golden/*.npz written by
oracle/make_golden.py from that reference.  Every function cites the reference
lines it follows (paths relative to /root/reference/).
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from oracle import detrand


@dataclass
class PortConfig:
  """The ``GlobalConfig`` attributes the hot path reads, with the reference defaults (team_code/config.py)."""
  camera_height: int = 256  # config.py:105
  camera_width: int = 1024  # config.py:104
  camera_fov: float = 110.0  # config.py:106
  camera_pos: tuple = (-1.5, 0.0, 2.0)  # config.py:100
  lidar_resolution_height: int = 256  # config.py:121
  lidar_resolution_width: int = 256  # config.py:120
  lidar_seq_len: int = 1  # config.py:115
  pixels_per_meter: float = 4.0  # config.py:126
  min_x: float = -32.0  # config.py:135-138
  max_x: float = 32.0
  min_y: float = -32.0
  max_y: float = 32.0
  min_z_projection: float = -10.0  # config.py:141-142
  max_z_projection: float = 14.0
  bev_grid_height_downsample_factor: float = 1.0
  img_vert_anchors: int = 8  # config.py:333-340
  img_horz_anchors: int = 32
  lidar_vert_anchors: int = 8
  lidar_horz_anchors: int = 8
  n_layer: int = 2  # config.py:351-357
  n_head: int = 4
  block_exp: int = 4
  embd_pdrop: float = 0.1
  resid_pdrop: float = 0.1
  attn_pdrop: float = 0.1
  bev_features_chanels: int = 64  # config.py:345-348
  bev_down_sample_factor: int = 4
  bev_upsample_factor: int = 2
  num_semantic_classes: int = 7
  num_bev_semantic_classes: int = 11
  deconv_channel_num_0: int = 128
  deconv_channel_num_1: int = 64
  deconv_channel_num_2: int = 32
  deconv_scale_factor_0: int = 4
  deconv_scale_factor_1: int = 8
  perspective_downsample_factor: int = 1
  gru_input_size: int = 256  # config.py:329-330
  gru_hidden_size: int = 64
  num_transformer_decoder_layers: int = 6  # config.py:468-469
  num_decoder_heads: int = 8
  decoder_dropout: float = 0.1  # nn.TransformerDecoderLayer default (model.py:137-140)
  # model.py:139 passes activation=nn.GELU(), but nn.TransformerDecoder deep-copies the layer and
  # nn.TransformerDecoderLayer.__setstate__ (torch 1.12.1 and 2.10 alike) then installs F.relu in the copy's
  # __dict__ because the GELU *module* lives in _modules, not in the pickled state -> every one of the 6 layers
  # the reference actually runs uses ReLU.  Verified against the reference in tests/test_oracle.py.
  decoder_activation: str = 'relu'
  predict_checkpoint_len: int = 10
  pred_len: int = 8  # config.py:118
  num_target_speeds: int = 4  # len(config.target_speeds), config.py:148
  num_bb_classes: int = 4
  num_dir_bins: int = 12
  use_wp_gru: bool = False  # config.py:370
  multi_wp_output: bool = False  # config.py:484: two waypoint hypotheses + a path-selection logit (with use_wp_gru)
  tp_attention: bool = False  # config.py:483: target point as a memory token of the reference's own attention-returning decoder (state_dict: generic_state_dict)
  use_controller_input_prediction: bool = True  # config.py:203
  target_speed_weights: tuple = (0.866605263873406, 7.4527377240841775, 1.2281629310898465, 0.5269622904065803)
  semantic_weights: tuple = (1.0,) * 7  # config.py:163
  bev_semantic_weights: tuple = (1.0,) * 11  # config.py:164
  regnet_widths: tuple = (72, 216, 576, 1512)  # timm regnety_032 (oracle/timm_regnet.py)
  regnet_depths: tuple = (2, 5, 13, 1)
  regnet_group_w: int = 24
  regnet_stem_w: int = 32
  extra: dict = field(default_factory=dict)


# ----------------------------------------------------------------------------------------------
# state_dict schema + deterministic initialisation
# ----------------------------------------------------------------------------------------------
def param_schema(cfg=None):
  """Ordered ``[(key, shape, kind)]`` of the reference's ``LidarCenterNet.state_dict()``.

  kind in {w (conv/linear weight), b (bias), g (norm gain), beta (norm shift), rm, rv, nbt, emb, mask, lossw}.
  Order follows module registration order in team_code/model.py:29-221 / transfuser.py:21-129."""
  cfg = cfg or PortConfig()
  out = []

  def add(k, shape, kind):
    out.append((k, tuple(shape), kind))

  def bn(p, c):
    add(p + '.weight', (c,), 'g')
    add(p + '.bias', (c,), 'beta')
    add(p + '.running_mean', (c,), 'rm')
    add(p + '.running_var', (c,), 'rv')
    add(p + '.num_batches_tracked', (), 'nbt')

  def convbn(p, cin, cout, k, groups=1):
    add(p + '.conv.weight', (cout, cin // groups, k, k), 'w')
    bn(p + '.bn', cout)

  def conv(p, cin, cout, k):
    add(p + '.weight', (cout, cin, k, k), 'w')
    add(p + '.bias', (cout,), 'b')

  def lin(p, cin, cout):
    add(p + '.weight', (cout, cin), 'w')
    add(p + '.bias', (cout,), 'b')

  def ln(p, c):
    add(p + '.weight', (c,), 'g')
    add(p + '.bias', (c,), 'beta')

  def regnet(p, in_ch):
    convbn(p + '.stem', in_ch, cfg.regnet_stem_w, 3)
    cin = cfg.regnet_stem_w
    for i, (w, d) in enumerate(zip(cfg.regnet_widths, cfg.regnet_depths)):
      for k in range(d):
        bp = f'{p}.s{i + 1}.b{k + 1}'
        convbn(bp + '.conv1', cin, w, 1)
        convbn(bp + '.conv2', w, w, 3, groups=w // cfg.regnet_group_w)
        rd = int(round(cin * 0.25))
        conv(bp + '.se.fc1', w, rd, 1)
        conv(bp + '.se.fc2', rd, w, 1)
        convbn(bp + '.conv3', w, w, 1)
        if k == 0:
          convbn(bp + '.downsample', cin, w, 1)
        cin = w

  # model.py:100-101 (frozen parameters registered after head + semantic decoder in __init__, but
  # nn.Module lists direct parameters before sub-modules)
  add('valid_bev_pixels', (1, 1, cfg.lidar_resolution_height, cfg.lidar_resolution_width), 'mask')
  add('valid_bev_pixels_inv', (1, 1, cfg.lidar_resolution_height, cfg.lidar_resolution_width), 'mask')
  add('extra_sensor_pos_embed', (1, cfg.gru_input_size), 'emb')
  if cfg.use_wp_gru:
    add('wp_query', (1, 2 * cfg.pred_len + 1 if cfg.multi_wp_output else cfg.pred_len, cfg.gru_input_size), 'emb')  # model.py:151-153,165-166
  if cfg.use_controller_input_prediction:
    add('checkpoint_query', (1, cfg.predict_checkpoint_len + 1, cfg.gru_input_size), 'emb')

  # backbone (transfuser.py:25-129)
  regnet('backbone.image_encoder', 3)
  regnet('backbone.lidar_encoder', cfg.lidar_seq_len)
  ntok = cfg.img_vert_anchors * cfg.img_horz_anchors + cfg.lidar_vert_anchors * cfg.lidar_horz_anchors
  for i, c in enumerate(cfg.regnet_widths):
    p = f'backbone.transformers.{i}'
    add(p + '.pos_emb', (1, ntok, c), 'emb')
    for l in range(cfg.n_layer):
      bp = f'{p}.blocks.{l}'
      ln(bp + '.ln1', c)
      ln(bp + '.ln2', c)
      for nme in ('key', 'query', 'value', 'proj'):
        lin(f'{bp}.attn.{nme}', c, c)
      lin(bp + '.mlp.0', c, cfg.block_exp * c)
      lin(bp + '.mlp.2', cfg.block_exp * c, c)
    ln(p + '.ln_f', c)
  for i, c in enumerate(cfg.regnet_widths):
    conv(f'backbone.lidar_channel_to_img.{i}', c, c, 1)
  for i, c in enumerate(cfg.regnet_widths):
    conv(f'backbone.img_channel_to_lidar.{i}', c, c, 1)
  ch = cfg.bev_features_chanels
  conv('backbone.up_conv5', ch, ch, 3)
  conv('backbone.up_conv4', ch, ch, 3)
  conv('backbone.c5_conv', cfg.regnet_widths[-1], ch, 1)

  # CenterNet head (center_net.py:23-31)
  for nme, k in (('heatmap', cfg.num_bb_classes), ('wh', 2), ('offset', 2), ('yaw_class', cfg.num_dir_bins),
                 ('yaw_res', 1)):
    conv(f'head.{nme}_head.0', ch, ch, 3)
    conv(f'head.{nme}_head.2', ch, k, 1)

  def persp(p, cout):  # transfuser_utils.py:674-695
    c0, c1, c2 = cfg.deconv_channel_num_0, cfg.deconv_channel_num_1, cfg.deconv_channel_num_2
    conv(p + '.deconv1.0', cfg.regnet_widths[-1], c0, 3)
    conv(p + '.deconv1.2', c0, c1, 3)
    conv(p + '.deconv2.0', c1, c2, 3)
    conv(p + '.deconv2.2', c2, c2, 3)
    conv(p + '.deconv3.0', c2, c2, 3)
    conv(p + '.deconv3.2', c2, cout, 3)

  persp('semantic_decoder', cfg.num_semantic_classes)
  conv('bev_semantic_decoder.0', ch, ch, 3)
  conv('bev_semantic_decoder.2', ch, cfg.num_bev_semantic_classes, 1)
  persp('depth_decoder', 1)
  d = cfg.gru_input_size
  if cfg.use_controller_input_prediction:
    lin('target_speed_network.0', d, d)
    lin('target_speed_network.2', d, cfg.num_target_speeds)
  for l in range(cfg.num_transformer_decoder_layers):  # nn.TransformerDecoderLayer parameter order
    p = f'join.layers.{l}'
    add(p + '.self_attn.in_proj_weight', (3 * d, d), 'w')
    add(p + '.self_attn.in_proj_bias', (3 * d,), 'b')
    lin(p + '.self_attn.out_proj', d, d)
    add(p + '.multihead_attn.in_proj_weight', (3 * d, d), 'w')
    add(p + '.multihead_attn.in_proj_bias', (3 * d,), 'b')
    lin(p + '.multihead_attn.out_proj', d, d)
    lin(p + '.linear1', d, 2048)
    lin(p + '.linear2', 2048, d)
    ln(p + '.norm1', d)
    ln(p + '.norm2', d)
    ln(p + '.norm3', d)
  ln('join.norm', d)
  conv('change_channel', cfg.regnet_widths[-1], d, 1)

  def gru_dec(p):  # model.py:848-855
    h = cfg.gru_hidden_size
    add(p + '.gru.weight_ih_l0', (3 * h, d), 'w')
    add(p + '.gru.weight_hh_l0', (3 * h, h), 'w')
    add(p + '.gru.bias_ih_l0', (3 * h,), 'b')
    add(p + '.gru.bias_hh_l0', (3 * h,), 'b')
    lin(p + '.encoder', 2, h)
    lin(p + '.decoder', h, 2)

  if cfg.use_wp_gru:
    gru_dec('wp_decoder')
    if cfg.multi_wp_output:  # model.py:159-163
      gru_dec('wp_decoder_1')
      lin('select_wps', d, 1)
  if cfg.use_controller_input_prediction:
    gru_dec('checkpoint_decoder')
  add('velocity_normalization.running_mean', (1,), 'rm')
  add('velocity_normalization.running_var', (1,), 'rv')
  add('velocity_normalization.num_batches_tracked', (), 'nbt')
  lin('extra_sensor_encoder.0', 7, 128)
  lin('extra_sensor_encoder.2', 128, d)
  # nn.CrossEntropyLoss(weight=...) registers its class weights as a buffer (model.py:260-265)
  add('loss_speed.weight', (cfg.num_target_speeds,), 'lossw')
  add('loss_semantic.weight', (cfg.num_semantic_classes,), 'lossw')
  add('loss_bev_semantic.weight', (cfg.num_bev_semantic_classes,), 'lossw')
  return out


def visibility_mask(cfg=None):
  """``valid_bev_pixels`` (1,1,H,W): which BEV pixels any voxel column projects into the camera.

  Follows team_code/transfuser_utils.py:596-665 (create_projection_grid: pinhole projection of voxel
  centres, un-rotated camera) and team_code/model.py:93-98 (max over height, transpose)."""
  cfg = cfg or PortConfig()
  mpp = 1.0 / cfg.pixels_per_meter
  widths = torch.arange(cfg.min_x, cfg.max_x, mpp) + mpp * 0.5
  depths = torch.arange(cfg.min_y, cfg.max_y, mpp) + mpp * 0.5
  mpph = mpp * cfg.bev_grid_height_downsample_factor
  heights = torch.arange(cfg.min_z_projection, cfg.max_z_projection, mpph) + mpph * 0.5
  dd, ww, hh = torch.meshgrid(depths, widths, heights, indexing='ij')
  cloud = torch.stack((dd, ww, hh), 0).reshape(3, -1) - torch.tensor(cfg.camera_pos).unsqueeze(1)
  cam = torch.stack((cloud[1], cloud[2], cloud[0]))  # x right, y down(z up as in reference), z front
  f = cfg.camera_width / (2.0 * np.tan(cfg.camera_fov * np.pi / 360.0))
  k = torch.from_numpy(np.array([[f, 0.0, cfg.camera_width / 2.0], [0.0, f, cfg.camera_height / 2.0],
                                 [0.0, 0.0, 1.0]])).to(torch.float32)
  proj = k @ cam
  z = proj[2:3]
  uv = proj[:2] / z
  ok = (uv[0:1] >= 0.0) & (uv[0:1] < cfg.camera_width) & (uv[1:2] >= 0.0) & (uv[1:2] < cfg.camera_height) & (z > 0.0)
  ok = ok.to(torch.float32).reshape(1, dd.shape[0], dd.shape[1], dd.shape[2])
  valid = ok.max(dim=3)[0].unsqueeze(1)
  return valid.transpose(2, 3).contiguous()


def make_state_dict(cfg=None, seed=0):
  """Deterministic *randomised* state_dict (oracle/detrand.py): every BN gain/shift/statistic, every
  bias and the zero-initialised ``pos_emb`` are non-trivial, so a broken kernel cannot hide behind
  timm's ``zero_init_last`` (SURVEY.md §7 "Hard parts")."""
  cfg = cfg or PortConfig()
  sd = {}
  mask = None
  for key, shape, kind in param_schema(cfg):
    if kind == 'w':
      fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
      a = math.sqrt(3.0 / fan_in)
      # last 1x1 of each bottleneck feeds a residual sum: keep the branch smaller so 21 blocks stay O(1)
      t = detrand.uniform(key, shape, -a, a, seed)
    elif kind == 'b':
      t = detrand.uniform(key, shape, -0.1, 0.1, seed)
    elif kind == 'g':
      lo, hi = (0.15, 0.45) if '.conv3.bn.' in key else (0.6, 1.4)
      t = detrand.uniform(key, shape, lo, hi, seed)
    elif kind == 'beta':
      t = detrand.uniform(key, shape, -0.1, 0.1, seed)
    elif kind == 'rm':
      t = detrand.uniform(key, shape, -0.2, 0.2, seed)
    elif kind == 'rv':
      t = detrand.uniform(key, shape, 0.6, 1.6, seed)
    elif kind == 'nbt':
      sd[key] = torch.tensor(0, dtype=torch.long)
      continue
    elif kind == 'emb':
      t = detrand.uniform(key, shape, 0.0, 1.0, seed) if 'query' in key or 'embed' in key else detrand.uniform(
          key, shape, -0.05, 0.05, seed)
    elif kind == 'mask':
      if mask is None:
        mask = visibility_mask(cfg)
      sd[key] = mask.clone() if key == 'valid_bev_pixels' else 1.0 - mask
      continue
    elif kind == 'lossw':
      src = {'loss_speed.weight': cfg.target_speed_weights, 'loss_semantic.weight': cfg.semantic_weights,
             'loss_bev_semantic.weight': cfg.bev_semantic_weights}[key]
      sd[key] = torch.tensor(src, dtype=torch.float32)
      continue
    else:
      raise KeyError(kind)
    sd[key] = torch.from_numpy(np.ascontiguousarray(t))
  sd['velocity_normalization.running_mean'] = torch.tensor([2.5])
  sd['velocity_normalization.running_var'] = torch.tensor([6.0])
  return sd


def generic_state_dict(model_sd, base=None, seed=0):
  """Deterministic randomised values for ANY state_dict schema (configurations beyond the default TransFuser++ one, e.g. the Video-Swin
  LiDAR branch of BASELINE config 5): keys present in ``base`` with the same shape keep their values, integer buffers are kept as they
  are, every other tensor is drawn from oracle/detrand.py by a rule on its name and shape."""
  out = {}
  for k, v in model_sd.items():
    shape = tuple(v.shape)
    if base is not None and k in base and tuple(base[k].shape) == shape:
      out[k] = base[k]
    elif not v.dtype.is_floating_point:
      out[k] = v.clone()
    elif k.split('.')[-1] in ('valid_bev_pixels', 'valid_bev_pixels_inv', 'grid', 'bev_projection_normalizer'):
      out[k] = v.clone()  # geometry constants registered as (frozen) parameters
    else:
      if k.endswith('relative_position_bias_table'):
        t = detrand.uniform(k, shape, -0.5, 0.5, seed)
      elif 'running_var' in k:
        t = detrand.uniform(k, shape, 0.6, 1.6, seed)
      elif len(shape) == 1 and k.endswith('.weight'):
        t = detrand.uniform(k, shape, 0.6, 1.4, seed)  # normalisation gains
      elif k.endswith('.bias') or len(shape) <= 1:
        t = detrand.uniform(k, shape, -0.1, 0.1, seed)
      elif 'pos_emb' in k:
        t = detrand.uniform(k, shape, -0.05, 0.05, seed)
      elif 'query' in k or 'embed' in k:
        t = detrand.uniform(k, shape, 0.0, 1.0, seed)
      else:
        a = math.sqrt(3.0 / int(np.prod(shape[1:])))
        t = detrand.uniform(k, shape, -a, a, seed)
      out[k] = torch.from_numpy(t).reshape(shape)
  return out


def make_inputs(batch, cfg=None, seed=1234):
  """Synthetic camera + LiDAR batch of SURVEY.md §8(d) (value ranges of team_code/train.py:750,
  team_code/data.py:887-889), generated with oracle/detrand.py."""
  cfg = cfg or PortConfig()
  b = batch
  rgb = detrand.randint('rgb', (b, 3, cfg.camera_height, cfg.camera_width), 0, 256, seed).astype(np.float32)
  occ = detrand.uniform01('lidar_occ', (b, cfg.lidar_seq_len, cfg.lidar_resolution_height, cfg.lidar_resolution_width),
                          seed) < 0.1
  cnt = detrand.randint('lidar_cnt', occ.shape, 1, 6, seed).astype(np.float32) / 5.0
  lidar = (occ * cnt).astype(np.float32)
  tp = detrand.uniform('target_point', (b, 2), -1.0, 1.0, seed) * np.array([[20.0, 5.0]], np.float32)
  vel = detrand.uniform('ego_vel', (b, 1), 0.0, 8.0, seed)
  cmd = np.eye(6, dtype=np.float32)[detrand.randint('command', (b,), 0, 6, seed)]
  return tuple(torch.from_numpy(np.ascontiguousarray(x)) for x in (rgb, lidar, tp.astype(np.float32), vel, cmd))


def make_labels(batch, cfg=None, seed=1234):
  """Synthetic training labels with the dtypes of team_code/train.py:693-766 and the CenterNet target
  conventions of team_code/data.py:722-791 (heat-map peaks are exactly 1.0)."""
  cfg = cfg or PortConfig()
  seed = cfg.extra.get('label_seed', seed)  # (a variant's fixture may ask for another draw: oracle/make_golden.py multi_wp)
  b = batch
  hb, wb = cfg.lidar_resolution_height // cfg.bev_down_sample_factor, cfg.lidar_resolution_width // cfg.bev_down_sample_factor
  lab = {}
  lab['target_speed_label'] = detrand.randint('ts', (b,), 0, cfg.num_target_speeds, seed)
  lab['checkpoint_label'] = detrand.uniform('route', (b, cfg.predict_checkpoint_len, 2), -10, 10, seed)
  lab['waypoint_label'] = detrand.uniform('wp', (b, cfg.pred_len, 2), -10, 10, seed)
  lab['semantic_label'] = detrand.randint('sem', (b, cfg.camera_height, cfg.camera_width), 0, cfg.num_semantic_classes,
                                          seed)
  lab['bev_semantic_label'] = detrand.randint('bevsem', (b, cfg.lidar_resolution_height, cfg.lidar_resolution_width), 0,
                                              cfg.num_bev_semantic_classes, seed)
  lab['depth_label'] = detrand.uniform('depth', (b, cfg.camera_height, cfg.camera_width), 0.0, 1.0, seed)
  heat = np.zeros((b, cfg.num_bb_classes, hb, wb), np.float32)
  pw = np.zeros((b, 2, hb, wb), np.float32)
  nbox = detrand.randint('nbox', (b,), 1, 6, seed)
  ys, xs = np.mgrid[0:hb, 0:wb]
  for i in range(b):
    cy = detrand.randint(f'cy{i}', (int(nbox[i]),), 4, hb - 4, seed)
    cx = detrand.randint(f'cx{i}', (int(nbox[i]),), 4, wb - 4, seed)
    cl = detrand.randint(f'cl{i}', (int(nbox[i]),), 0, cfg.num_bb_classes, seed)
    for y, x, c in zip(cy, cx, cl):
      g = np.exp(-((ys - y)**2 + (xs - x)**2) / (2 * 1.5**2)).astype(np.float32)
      g[y, x] = 1.0
      heat[i, c] = np.maximum(heat[i, c], g)
      pw[i, :, y, x] = 1.0
  lab['center_heatmap_label'] = heat
  lab['wh_label'] = detrand.uniform('wh', (b, 2, hb, wb), 0.0, 8.0, seed)
  lab['yaw_class_label'] = detrand.randint('yawc', (b, hb, wb), 0, cfg.num_dir_bins, seed)
  lab['yaw_res_label'] = detrand.uniform('yawr', (b, 1, hb, wb), -0.3, 0.3, seed)
  lab['offset_label'] = detrand.uniform('off', (b, 2, hb, wb), 0.0, 1.0, seed)
  lab['velocity_label'] = detrand.uniform('vel', (b, 1, hb, wb), 0.0, 8.0, seed)
  lab['brake_target_label'] = detrand.randint('brk', (b, hb, wb), 0, 2, seed)
  lab['pixel_weight_label'] = pw
  lab['avg_factor_label'] = nbox.astype(np.float32)
  return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in lab.items()}


# ----------------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------------
def _bn(x, sd, p, training):
  return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'], training,
                      0.1, 1e-5)


def _convbn(x, sd, p, stride=1, groups=1, act=True, training=False):
  w = sd[p + '.conv.weight']
  x = _bn(F.conv2d(x, w, None, stride, w.shape[-1] // 2, 1, groups), sd, p + '.bn', training)
  return F.relu(x) if act else x


def _bottleneck(x, sd, p, stride, group_w, training):
  """RegNet-Y block, oracle/timm_regnet.py::_Bottleneck (timm 0.6.7 regnet.Bottleneck)."""
  sc = _convbn(x, sd, p + '.downsample', stride, act=False, training=training) if (p + '.downsample.conv.weight'
                                                                                   in sd) else x
  y = _convbn(x, sd, p + '.conv1', training=training)
  y = _convbn(y, sd, p + '.conv2', stride, y.shape[1] // group_w, training=training)
  s = y.mean((2, 3), keepdim=True)
  s = F.conv2d(F.relu(F.conv2d(s, sd[p + '.se.fc1.weight'], sd[p + '.se.fc1.bias'])), sd[p + '.se.fc2.weight'],
               sd[p + '.se.fc2.bias'])
  y = y * torch.sigmoid(s)
  y = _convbn(y, sd, p + '.conv3', act=False, training=training)
  return F.relu(y + sc)


def _stage(x, sd, p, depth, cfg, training):
  for k in range(depth):
    x = _bottleneck(x, sd, f'{p}.b{k + 1}', 2 if k == 0 else 1, cfg.regnet_group_w, training)
  return x


def _lin(x, sd, p):
  return F.linear(x, sd[p + '.weight'], sd[p + '.bias'])


def _ln(x, sd, p):
  return F.layer_norm(x, (x.shape[-1],), sd[p + '.weight'], sd[p + '.bias'], 1e-5)


def _gpt(img, lid, sd, p, cfg, training):
  """team_code/transfuser.py:301-339 (GPT.forward), 383-402 (Block), 362-380 (SelfAttention)."""
  b, c, ih, iw = img.shape
  lh, lw = lid.shape[2:]
  tok = torch.cat((img.permute(0, 2, 3, 1).reshape(b, -1, c), lid.permute(0, 2, 3, 1).reshape(b, -1, c)), 1)
  x = F.dropout(sd[p + '.pos_emb'] + tok, cfg.embd_pdrop, training)
  t = x.shape[1]
  nh = cfg.n_head
  for l in range(cfg.n_layer):
    bp = f'{p}.blocks.{l}'
    h = _ln(x, sd, bp + '.ln1')
    k = _lin(h, sd, bp + '.attn.key').view(b, t, nh, c // nh).transpose(1, 2)
    q = _lin(h, sd, bp + '.attn.query').view(b, t, nh, c // nh).transpose(1, 2)
    v = _lin(h, sd, bp + '.attn.value').view(b, t, nh, c // nh).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(c // nh))
    att = F.dropout(F.softmax(att, -1), cfg.attn_pdrop, training)
    y = (att @ v).transpose(1, 2).reshape(b, t, c)
    x = x + F.dropout(_lin(y, sd, bp + '.attn.proj'), cfg.resid_pdrop, training)
    h = _ln(x, sd, bp + '.ln2')
    h = _lin(F.relu(_lin(h, sd, bp + '.mlp.0')), sd, bp + '.mlp.2')
    x = x + F.dropout(h, cfg.resid_pdrop, training)
  x = _ln(x, sd, p + '.ln_f')
  ni = ih * iw
  img_o = x[:, :ni].reshape(b, ih, iw, c).permute(0, 3, 1, 2)
  lid_o = x[:, ni:].reshape(b, lh, lw, c).permute(0, 3, 1, 2)
  return img_o, lid_o


def normalize_imagenet(x):
  """team_code/transfuser_utils.py:542-551."""
  mean = x.new_tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
  std = x.new_tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
  return ((x / 255.0) - mean) / std


def backbone_forward(sd, cfg, image, lidar, training=False, taps=None):
  """team_code/transfuser.py:139-205 (TransfuserBackbone.forward) + 222-257 (fuse_features) + 131-137 (top_down)."""
  pi, pl = 'backbone.image_encoder', 'backbone.lidar_encoder'
  xi = _convbn(normalize_imagenet(image), sd, pi + '.stem', 2, training=training)
  xl = _convbn(lidar, sd, pl + '.stem', 2, training=training)
  if taps is not None:
    taps['img_stem'] = xi
    taps['lidar_stem'] = xl
  for i, d in enumerate(cfg.regnet_depths):
    xi = _stage(xi, sd, f'{pi}.s{i + 1}', d, cfg, training)
    xl = _stage(xl, sd, f'{pl}.s{i + 1}', d, cfg, training)
    if taps is not None:
      taps[f'img_s{i + 1}'] = xi
      taps[f'lidar_s{i + 1}'] = xl
    ie = F.adaptive_avg_pool2d(xi, (cfg.img_vert_anchors, cfg.img_horz_anchors))
    le = F.adaptive_avg_pool2d(xl, (cfg.lidar_vert_anchors, cfg.lidar_horz_anchors))
    le = F.conv2d(le, sd[f'backbone.lidar_channel_to_img.{i}.weight'], sd[f'backbone.lidar_channel_to_img.{i}.bias'])
    io, lo = _gpt(ie, le, sd, f'backbone.transformers.{i}', cfg, training)
    lo = F.conv2d(lo, sd[f'backbone.img_channel_to_lidar.{i}.weight'], sd[f'backbone.img_channel_to_lidar.{i}.bias'])
    xi = xi + F.interpolate(io, size=xi.shape[2:], mode='bilinear', align_corners=False)
    xl = xl + F.interpolate(lo, size=xl.shape[2:], mode='bilinear', align_corners=False)
    if taps is not None:
      taps[f'img_f{i + 1}'] = xi
      taps[f'lidar_f{i + 1}'] = xl
  p5 = F.relu(F.conv2d(xl, sd['backbone.c5_conv.weight'], sd['backbone.c5_conv.bias']))
  p4 = F.interpolate(p5, scale_factor=cfg.bev_upsample_factor, mode='bilinear', align_corners=False)
  p4 = F.relu(F.conv2d(p4, sd['backbone.up_conv5.weight'], sd['backbone.up_conv5.bias'], padding=1))
  p3 = F.interpolate(p4,
                     size=(cfg.lidar_resolution_height // cfg.bev_down_sample_factor,
                           cfg.lidar_resolution_width // cfg.bev_down_sample_factor),
                     mode='bilinear',
                     align_corners=False)
  p3 = F.relu(F.conv2d(p3, sd['backbone.up_conv4.weight'], sd['backbone.up_conv4.bias'], padding=1))
  return p3, xl, xi


def position_embedding_sine(h, w, num_pos_feats, temperature=10000.0):
  """team_code/model.py:934-953 with normalize=True, scale=2*pi: (1, 2*num_pos_feats, h, w) constant."""
  y = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w)
  x = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w)
  y = y / (float(h) + 1e-6) * (2 * math.pi)
  x = x / (float(w) + 1e-6) * (2 * math.pi)
  dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
  dim_t = temperature**(2 * torch.div(dim_t, 2, rounding_mode='floor') / num_pos_feats)
  px = x[:, :, None] / dim_t
  py = y[:, :, None] / dim_t
  px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), 3).flatten(2)
  py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), 3).flatten(2)
  return torch.cat((py, px), 2).permute(2, 0, 1).unsqueeze(0)


def _mha(q_in, kv_in, sd, p, nh, pdrop, training):
  """torch.nn.MultiheadAttention math (batch_first) as used by nn.TransformerDecoderLayer (model.py:137-143)."""
  b, tq, d = q_in.shape
  tk = kv_in.shape[1]
  w, bias = sd[p + '.in_proj_weight'], sd[p + '.in_proj_bias']
  q = F.linear(q_in, w[:d], bias[:d]).view(b, tq, nh, d // nh).transpose(1, 2)
  k = F.linear(kv_in, w[d:2 * d], bias[d:2 * d]).view(b, tk, nh, d // nh).transpose(1, 2)
  v = F.linear(kv_in, w[2 * d:], bias[2 * d:]).view(b, tk, nh, d // nh).transpose(1, 2)
  att = F.softmax((q @ k.transpose(-2, -1)) / math.sqrt(d // nh), -1)
  att = F.dropout(att, pdrop, training)
  y = (att @ v).transpose(1, 2).reshape(b, tq, d)
  return _lin(y, sd, p + '.out_proj')


def _decoder(query, memory, sd, cfg, training):
  """nn.TransformerDecoder, post-norm layers + final LayerNorm (model.py:137-143,352).  FFN activation:
  see PortConfig.decoder_activation (ReLU as actually run, not the GELU the source asks for)."""
  act = F.relu if cfg.decoder_activation == 'relu' else F.gelu
  x = query
  pd = cfg.decoder_dropout
  for l in range(cfg.num_transformer_decoder_layers):
    p = f'join.layers.{l}'
    x = _ln(x + F.dropout(_mha(x, x, sd, p + '.self_attn', cfg.num_decoder_heads, pd, training), pd, training), sd,
            p + '.norm1')
    x = _ln(x + F.dropout(_mha(x, memory, sd, p + '.multihead_attn', cfg.num_decoder_heads, pd, training), pd, training),
            sd, p + '.norm2')
    h = _lin(F.dropout(act(_lin(x, sd, p + '.linear1')), pd, training), sd, p + '.linear2')
    x = _ln(x + F.dropout(h, pd, training), sd, p + '.norm3')
  return _ln(x, sd, 'join.norm')


def _attn_with_weights(q_in, kv_in, sd, p, nh, pdrop, training):
  """team_code/transfuser.py:404-443 (MultiheadAttentionWithAttention): separate key / query / value / proj linears, dropout on the probabilities and
  on the projected output; returns (y, probabilities averaged over the heads)."""
  b, t, c = q_in.shape
  tm = kv_in.shape[1]
  q = _lin(q_in, sd, p + '.query').view(b, t, nh, c // nh).transpose(1, 2)
  k = _lin(kv_in, sd, p + '.key').view(b, tm, nh, c // nh).transpose(1, 2)
  v = _lin(kv_in, sd, p + '.value').view(b, tm, nh, c // nh).transpose(1, 2)
  att = F.dropout(F.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(c // nh)), dim=-1), pdrop, training)
  y = (att @ v).transpose(1, 2).contiguous().view(b, t, c)
  return F.dropout(_lin(y, sd, p + '.proj'), pdrop, training), att.mean(dim=1)


def _decoder_with_attention(query, memory, sd, cfg, training):
  """team_code/transfuser.py:447-508 (TransformerDecoderLayerWithAttention x num_layers + final norm), the decoder of config.tp_attention: post-norm
  layers, an exact GELU in the FFN (the activation module survives the deep copy here, unlike nn.TransformerDecoderLayer's), returns
  (output, cross-attention probabilities averaged over heads and layers)."""
  x, pd, atts = query, cfg.decoder_dropout, []
  for l in range(cfg.num_transformer_decoder_layers):
    p = f'join.layers.{l}'
    a, _ = _attn_with_weights(x, x, sd, p + '.self_attn', cfg.num_decoder_heads, pd, training)
    x = _ln(x + F.dropout(a, pd, training), sd, p + '.norm1')
    a, att = _attn_with_weights(x, memory, sd, p + '.multihead_attn', cfg.num_decoder_heads, pd, training)
    atts.append(att)
    x = _ln(x + F.dropout(a, pd, training), sd, p + '.norm2')
    h = _lin(F.dropout(F.gelu(_lin(x, sd, p + '.linear1')), pd, training), sd, p + '.linear2')
    x = _ln(x + F.dropout(h, pd, training), sd, p + '.norm3')
  return _ln(x, sd, 'join.norm'), torch.stack(atts).mean(dim=0)


def _gru_decoder(x, target_point, sd, p):
  """team_code/model.py:857-867 (GRUWaypointsPredictorInterFuser.forward); torch.nn.GRU gate equations."""
  h = _lin(target_point, sd, p + '.encoder')
  wi, wh = sd[p + '.gru.weight_ih_l0'], sd[p + '.gru.weight_hh_l0']
  bi, bh = sd[p + '.gru.bias_ih_l0'], sd[p + '.gru.bias_hh_l0']
  hs = h.shape[1]
  outs = []
  for t in range(x.shape[1]):
    gi = F.linear(x[:, t], wi, bi)
    gh = F.linear(h, wh, bh)
    r = torch.sigmoid(gi[:, :hs] + gh[:, :hs])
    z = torch.sigmoid(gi[:, hs:2 * hs] + gh[:, hs:2 * hs])
    n = torch.tanh(gi[:, 2 * hs:] + r * gh[:, 2 * hs:])
    h = (1.0 - z) * n + z * h
    outs.append(_lin(h, sd, p + '.decoder'))
  return torch.cumsum(torch.stack(outs, 1), 1)


def _perspective_decoder(x, sd, p, cfg):
  """team_code/transfuser_utils.py:697-704; scale factors from team_code/model.py:71-72."""
  up = 32 // cfg.perspective_downsample_factor
  s0, s1 = up // cfg.deconv_scale_factor_0, up // cfg.deconv_scale_factor_1

  def c(x, name, act=True):
    x = F.conv2d(x, sd[f'{p}.{name}.weight'], sd[f'{p}.{name}.bias'], padding=1)
    return F.relu(x) if act else x

  x = c(c(x, 'deconv1.0'), 'deconv1.2')
  x = F.interpolate(x, scale_factor=s0, mode='bilinear', align_corners=False)
  x = c(c(x, 'deconv2.0'), 'deconv2.2')
  x = F.interpolate(x, scale_factor=s1, mode='bilinear', align_corners=False)
  return c(c(x, 'deconv3.0'), 'deconv3.2', act=False)


def _head_branch(x, sd, p):
  x = F.relu(F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'], padding=1))
  return F.conv2d(x, sd[p + '.2.weight'], sd[p + '.2.bias'])


def forward(sd, cfg, rgb, lidar_bev, target_point, ego_vel, command, training=False, taps=None):
  """team_code/model.py:279-392 (LidarCenterNet.forward) -> the reference's 10-tuple."""
  bs = rgb.shape[0]
  bev, fused, img_grid = backbone_forward(sd, cfg, rgb, lidar_bev, training, taps)
  pred_wp = pred_ts = pred_cp = None
  x = F.conv2d(fused, sd['change_channel.weight'], sd['change_channel.bias'])  # model.py:301
  x = x + position_embedding_sine(x.shape[2], x.shape[3], cfg.gru_input_size // 2).to(x)  # model.py:302
  x = torch.flatten(x, 2)
  vel = F.batch_norm(ego_vel, sd['velocity_normalization.running_mean'], sd['velocity_normalization.running_var'],
                     None, None, training, 0.1, 1e-5)  # model.py:216,311
  es = torch.cat((vel, command), 1)
  es = F.relu(_lin(F.relu(_lin(es, sd, 'extra_sensor_encoder.0')), sd, 'extra_sensor_encoder.2'))  # model.py:220,315
  es = es + sd['extra_sensor_pos_embed'].repeat(bs, 1)
  mem = torch.cat((x, es.unsqueeze(2)), 2).permute(0, 2, 1)  # model.py:319,324
  if taps is not None:
    taps['memory'] = mem
  pred_wp_1 = selected_path = None
  if cfg.use_wp_gru and cfg.multi_wp_output:  # model.py:326-331
    j = _decoder(sd['wp_query'].repeat(bs, 1, 1), mem, sd, cfg, training)
    n = cfg.pred_len
    pred_wp = _gru_decoder(j[:, :n], target_point, sd, 'wp_decoder')
    pred_wp_1 = _gru_decoder(j[:, n:2 * n], target_point, sd, 'wp_decoder_1')
    selected_path = _lin(j[:, 2 * n], sd, 'select_wps')
  elif cfg.use_wp_gru:
    j = _decoder(sd['wp_query'].repeat(bs, 1, 1), mem, sd, cfg, training)
    pred_wp = _gru_decoder(j, target_point, sd, 'wp_decoder')  # model.py:333-334
  attention_weights = None
  if cfg.use_controller_input_prediction and cfg.tp_attention:  # model.py:336-350
    tp_token = _lin(F.relu(_lin(target_point, sd, 'tp_encoder.0')), sd, 'tp_encoder.2') + sd['tp_pos_embed']
    npix = mem.shape[1] - 1
    j, att = _decoder_with_attention(sd['checkpoint_query'].repeat(bs, 1, 1), torch.cat((mem, tp_token.unsqueeze(1)), 1), sd, cfg, training)
    ga = att[:, :cfg.predict_checkpoint_len].mean(dim=1)[0]
    attention_weights = [ga[:npix].sum().item(), ga[npix].item(), ga[npix + 1].item()]
  elif cfg.use_controller_input_prediction:
    j = _decoder(sd['checkpoint_query'].repeat(bs, 1, 1), mem, sd, cfg, training)  # model.py:352
  if cfg.use_controller_input_prediction:
    if taps is not None:
      taps['joined'] = j
    n = cfg.predict_checkpoint_len
    pred_cp = _gru_decoder(j[:, :n], target_point, sd, 'checkpoint_decoder')  # model.py:354,357
    pred_ts = _lin(F.relu(_lin(j[:, n], sd, 'target_speed_network.0')), sd, 'target_speed_network.2')  # 355,358
  pred_sem = _perspective_decoder(img_grid, sd, 'semantic_decoder', cfg)  # model.py:373-374
  pred_depth = torch.sigmoid(_perspective_decoder(img_grid, sd, 'depth_decoder', cfg)).squeeze(1)  # 377-379
  y = F.relu(F.conv2d(bev, sd['bev_semantic_decoder.0.weight'], sd['bev_semantic_decoder.0.bias'], padding=1))
  y = F.conv2d(y, sd['bev_semantic_decoder.2.weight'], sd['bev_semantic_decoder.2.bias'])
  y = F.interpolate(y, size=(cfg.lidar_resolution_height, cfg.lidar_resolution_width), mode='bilinear',
                    align_corners=False)
  pred_bev = y * sd['valid_bev_pixels']  # model.py:383-385
  bb = (torch.sigmoid(_head_branch(bev, sd, 'head.heatmap_head')), _head_branch(bev, sd, 'head.wh_head'),
        _head_branch(bev, sd, 'head.offset_head'), _head_branch(bev, sd, 'head.yaw_class_head'),
        _head_branch(bev, sd, 'head.yaw_res_head'), None, None)  # center_net.py:49-75
  return pred_wp, pred_ts, pred_cp, pred_sem, pred_bev, pred_depth, bb, attention_weights, pred_wp_1, selected_path


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def gaussian_focal_loss_sum(pred, target, alpha=2.0, gamma=4.0):
  """team_code/transfuser_utils.py:341-364, reduction='sum'."""
  eps = 1e-12
  pos = target.eq(1)
  neg_w = (1 - target).pow(gamma)
  pos_loss = -(pred + eps).log() * (1 - pred).pow(alpha) * pos
  neg_loss = -(1 - pred + eps).log() * pred.pow(alpha) * neg_w
  return (pos_loss + neg_loss).sum()


def compute_loss(sd, cfg, outputs, labels):
  """team_code/model.py:394-445 + team_code/center_net.py:77-123 -> dict of 0-d losses (default heads)."""
  pred_wp, pred_ts, pred_cp, pred_sem, pred_bev, pred_depth, bb = outputs[:7]
  loss = {}
  if cfg.use_wp_gru and cfg.multi_wp_output:  # model.py:401-411
    per = torch.stack([torch.mean(torch.abs(w - labels['waypoint_label']), dim=(1, 2)) for w in (pred_wp, outputs[8])], dim=1)
    best, pick = torch.min(per, dim=1, keepdim=True)
    loss['loss_wp'] = torch.mean(best)
    loss['loss_selection'] = F.binary_cross_entropy_with_logits(outputs[9], pick.detach().float())
  elif cfg.use_wp_gru:
    loss['loss_wp'] = torch.mean(torch.abs(pred_wp - labels['waypoint_label']))
  if cfg.use_controller_input_prediction:
    loss['loss_target_speed'] = F.cross_entropy(pred_ts, labels['target_speed_label'],
                                                weight=pred_ts.new_tensor(cfg.target_speed_weights))
    loss['loss_checkpoint'] = torch.mean(torch.abs(pred_cp - labels['checkpoint_label']))
  loss['loss_semantic'] = F.cross_entropy(pred_sem, labels['semantic_label'],
                                          weight=pred_sem.new_tensor(cfg.semantic_weights))
  vis = sd['valid_bev_pixels'].squeeze(1).int()
  bev_lab = (vis - 1) + vis * labels['bev_semantic_label']  # model.py:427-429
  loss['loss_bev_semantic'] = F.cross_entropy(pred_bev, bev_lab.long(), weight=pred_bev.new_tensor(
      cfg.bev_semantic_weights), ignore_index=-1)
  loss['loss_depth'] = F.l1_loss(pred_depth, labels['depth_label'])
  pw = labels['pixel_weight_label']
  af = labels['avg_factor_label'].sum() + torch.finfo(torch.float32).eps
  loss['loss_center_heatmap'] = gaussian_focal_loss_sum(bb[0], labels['center_heatmap_label']) / af
  loss['loss_wh'] = (torch.abs(bb[1] - labels['wh_label']) * pw).sum() / (af * bb[1].shape[1])
  loss['loss_offset'] = (torch.abs(bb[2] - labels['offset_label']) * pw).sum() / (af * bb[1].shape[1])
  loss['loss_yaw_class'] = (F.cross_entropy(bb[3], labels['yaw_class_label'], reduction='none') * pw[:, 0]).sum() / af
  loss['loss_yaw_res'] = (F.smooth_l1_loss(bb[4], labels['yaw_res_label'], reduction='none') * pw[:, 0:1]).sum() / af
  return loss


def loss_weights(cfg):
  """Normalised per-loss weights as team_code/train.py:383-456 derives them from
  ``detailed_loss_weights`` (config.py:223-239): unused losses zeroed, the rest divided by their sum."""
  w = {
      'loss_wp': 1.0 if cfg.use_wp_gru else 0.0,
      'loss_selection': 1.0 if (cfg.use_wp_gru and cfg.multi_wp_output) else 0.0,  # train.py:440-441
      'loss_target_speed': 1.0 if cfg.use_controller_input_prediction else 0.0,
      'loss_checkpoint': 1.0 if cfg.use_controller_input_prediction else 0.0,
      'loss_semantic': 1.0,
      'loss_bev_semantic': 1.0,
      'loss_depth': 1.0,
      'loss_center_heatmap': 1.0,
      'loss_wh': 1.0,
      'loss_offset': 1.0,
      'loss_yaw_class': 1.0,
      'loss_yaw_res': 1.0,
  }
  if not (cfg.lidar_seq_len == 1 and getattr(cfg, 'seq_len', 1) == 1):  # train.py:423-426 zeroes them for single-frame input only
    w['loss_velocity'] = w['loss_brake'] = 1.0
  s = sum(w.values())
  return {k: v / s for k, v in w.items()}


def total_loss(sd, cfg, outputs, labels):
  """team_code/train.py:889-896: weighted sum of the individual losses."""
  losses = compute_loss(sd, cfg, outputs, labels)
  w = loss_weights(cfg)
  return sum(w[k] * v for k, v in losses.items()), losses
