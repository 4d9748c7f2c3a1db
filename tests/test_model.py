"""Boundary-module tests.  CPU part (-m "not gpu"): state_dict schema / init / optimizer groups / ABI exports.
GPU part (-m gpu): forward parity with the reference golden vectors and the CPU oracle (north_star: 1e-3 relative,
fp32), training-step parity (losses, gradients, BN statistics) and the drop-in autograd path."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

import parity_util as U
from oracle import tfpp_port as P
from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.model import LidarCenterNet
from carla_garage_amd import _lib


@pytest.fixture(scope='module')
def model_cpu():
  return LidarCenterNet(GlobalConfig())


def test_state_dict_schema_matches_reference(model_cpu):
  with open(os.path.join(U.GOLDEN, 'state_dict_schema.json'), encoding='utf-8') as f:
    gold = json.load(f)
  sd = model_cpu.state_dict()
  assert list(sd.keys()) == [e[0] for e in gold['entries']]
  assert [list(v.shape) for v in sd.values()] == [e[1] for e in gold['entries']]
  assert [str(v.dtype).replace('torch.', '') for v in sd.values()] == [e[2] for e in gold['entries']]
  assert sum(p.numel() for p in model_cpu.parameters() if p.requires_grad) == gold['n_trainable'] == 120219954
  torch.testing.assert_close(model_cpu.valid_bev_pixels.data, P.visibility_mask(P.PortConfig()))
  model_cpu.load_state_dict(P.make_state_dict(), strict=True)


def test_init_follows_reference_conventions():
  m = LidarCenterNet(GlobalConfig())
  blk = m.backbone.image_encoder['s3'].b2
  assert float(blk.conv3.bn.weight.abs().max()) == 0.0  # timm zero_init_last
  assert float(m.backbone.transformers[0].pos_emb.abs().max()) == 0.0
  w = m.backbone.transformers[3].blocks[0].mlp[0].weight
  assert abs(float(w.std()) - 0.02) < 2e-3
  assert 0.0 <= float(m.checkpoint_query.min()) and float(m.checkpoint_query.max()) <= 1.0
  assert all(l.activation is torch.nn.functional.relu for l in m.join.layers)  # as the reference actually runs
  torch.testing.assert_close(m.sine_table(8, 8).t().reshape(1, 256, 8, 8), P.position_embedding_sine(8, 8, 128))


def test_optimizer_groups_cover_every_parameter(model_cpu):
  groups = model_cpu.create_optimizer_groups(0.01)
  n = sum(len(g['params']) for g in groups)
  assert n == len(list(model_cpu.parameters()))
  names = {id(p): k for k, p in model_cpu.named_parameters()}
  decay = {names[id(p)] for p in groups[0]['params']}
  assert 'backbone.image_encoder.s1.b1.conv1.conv.weight' in decay and 'change_channel.weight' in decay
  assert 'checkpoint_decoder.gru.weight_ih_l0' in decay
  assert not any(k.endswith('bias') or '.bn.' in k or '.ln' in k or 'norm' in k for k in decay)


def test_optimizer_groups_of_a_syncbatchnorm_converted_model_equal_the_unconverted_ones():
  """train.py:511-512 converts the BatchNorm layers BEFORE train.py:523 builds the optimizer groups; nn.SyncBatchNorm is no subclass of
  BatchNorm2d / BatchNorm1d, and the reference's name rules keep the converted layers' weights in the no-decay group."""
  import copy
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  torch.manual_seed(0)
  plain = LidarCenterNet(GlobalConfig())
  conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(plain))
  assert any(isinstance(mod, torch.nn.SyncBatchNorm) for mod in conv.modules())
  for model_a, model_b in ((plain, conv),):
    na = {id(p): k for k, p in model_a.named_parameters()}
    nb = {id(p): k for k, p in model_b.named_parameters()}
    for ga, gb in zip(model_a.create_optimizer_groups(0.01), model_b.create_optimizer_groups(0.01)):
      assert ga['weight_decay'] == gb['weight_decay']
      assert sorted(na[id(p)] for p in ga['params']) == sorted(nb[id(p)] for p in gb['params'])


def test_library_exports_every_declared_symbol():
  _lib.build()
  decl = _lib.declared_functions()
  assert len(decl) >= 40
  _lib.lib.load()  # raises if a declared symbol is missing or a struct mirror has the wrong size


def test_abi_version_of_header_and_binding_agree():
  """include/tfpp.h is the contract: a library built from another header revision must not load (bump on every struct change)."""
  import re
  hdr = open(os.path.join(os.path.dirname(U.GOLDEN), '..', 'include', 'tfpp.h'), encoding='utf-8').read()
  v = int(re.search(r'#define\s+TFPP_ABI_VERSION\s+(\d+)', hdr).group(1))
  assert v == _lib.ABI_VERSION
  _lib.build()
  assert _lib.lib.raw('tfpp_version')() == v


def test_sync_batchnorm_conversion_is_accepted_and_marks_the_engine(model_cpu):
  """train.py:511-512 converts the module with nn.SyncBatchNorm.convert_sync_batchnorm when config.sync_batch_norm = 1 (default 0).  Round 5: the
  converted module runs (BatchNorm statistics all-reduced over the ranks, tests/test_dist_gpu.py::test_two_ranks_one_gpu_sync_batchnorm_*); its
  engine knows it and such a training step is never captured into a hipGraph."""
  import copy
  m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(model_cpu))
  assert isinstance(m.velocity_normalization, torch.nn.SyncBatchNorm) and isinstance(m.backbone.image_encoder['s3'].b2.conv1.bn, torch.nn.SyncBatchNorm)
  m.__dict__['engine'] = None
  eng = m._engine()
  assert eng.sync_bn and eng.sync_group is None and eng.sync_world == 1
  assert list(m.state_dict().keys()) == list(model_cpu.state_dict().keys())  # the checkpoint schema is unchanged by the conversion
  plain = copy.deepcopy(model_cpu)
  plain.__dict__['engine'] = None
  assert not plain._engine().sync_bn


AIM_CFG = dict(backbone='aim', use_semantic=0, use_depth=0, detect_boxes=0, use_bev_semantic=0)  # BASELINE config 1


def test_aim_state_dict_schema_matches_reference():
  """BASELINE config 1 (image-only AIM backbone, team_code/aim.py): same keys in the same order as the reference model."""
  g = U.load_golden('tfpp_aim.npz')
  m = LidarCenterNet(GlobalConfig(**AIM_CFG))
  assert list(m.state_dict().keys()) == [str(k) for k in g['keys']]
  full = P.make_state_dict()
  m.load_state_dict({k: full[k] for k in m.state_dict().keys()}, strict=True)
  with pytest.raises(ValueError):
    LidarCenterNet(GlobalConfig(backbone='aim'))  # the reference cannot build the dense heads on AIM either (model.py:71)


def test_every_library_call_of_the_host_code_is_declared_in_the_header():
  """The ctypes binding is generated from include/tfpp.h: a call of an undeclared entry point would only fail on the GPU box."""
  import glob
  import re
  decl = set(_lib.declared_functions())
  used = set()
  for path in glob.glob(os.path.join(os.path.dirname(_lib.__file__), '*.py')) + [os.path.join(_lib.ROOT, 'bench.py')]:
    with open(path, encoding='utf-8') as f:
      text = f.read()
    used |= set(re.findall(r"lib\.(tfpp_\w+)\(", text)) | set(re.findall(r"raw\('(tfpp_\w+)'\)", text))
  used.discard('tfpp_xxx')  # the docstring of _lib.py
  assert len(used) > 50 and not (used - decl), sorted(used - decl)


def test_cpu_tensors_are_rejected_not_silently_computed(model_cpu):
  inp = P.make_inputs(1)
  with pytest.raises(RuntimeError):
    model_cpu(*inp)


# ------------------------------------------------------------------------------------------------------------- GPU
def _model(dtype='fp32', **over):
  m = LidarCenterNet(GlobalConfig(tfpp_dtype=dtype, **over))
  m.load_state_dict(P.make_state_dict(), strict=True)
  return m.cuda()


def _report(name, errs):
  try:
    path = os.path.join(os.path.dirname(U.GOLDEN), '..', 'gpurun_out', 'model_report.jsonl')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'a', encoding='utf-8') as f:
      f.write(json.dumps({'test': name, 'errs': errs}) + '\n')
  except OSError:
    pass


def test_arena_layout_static_guess_and_observed_buckets(model_cpu):
  """Flat-arena layout used for the overlapped gradient exchange (engine.arena_layout): every trainable parameter exactly once, buckets
  contiguous and starting on 128-element boundaries.  Before a backward pass has been observed the static guess gives two buckets (heads,
  decoders, planning head and stage 4 of the backbone first: the larger part of the 481 MB); with an observed assignment
  (model._grad_buckets, set by Trainer.apply_observed_layout) the buckets follow it."""
  from carla_garage_amd.engine import arena_layout, finishes_early, BUCKET_ALIGN
  pad4 = lambda k: (k + 3) // 4 * 4
  want = [n for n, p in model_cpu.named_parameters() if p.requires_grad]

  def check(layout, total, offsets, key):
    names = [n for n, _, _ in layout]
    assert sorted(names) == sorted(want) and len(set(names)) == len(names)
    assert offsets[0] == 0 and offsets[-1] == total and all(o % BUCKET_ALIGN == 0 for o in offsets[:-1])
    prev_end = 0
    for n, p, off in layout:
      b = key(n)
      assert offsets[b] <= off and off + p.numel() <= offsets[b + 1], (n, b, off)   # inside its bucket
      assert off >= prev_end and off % 4 == 0                                        # no overlap, 16-byte aligned slots
      prev_end = off + pad4(p.numel())

  model_cpu.__dict__.pop('_grad_buckets', None)
  layout, total, offsets = arena_layout(model_cpu)
  assert len(offsets) == 3
  check(layout, total, offsets, lambda n: 0 if finishes_early(n) else 1)
  assert 0.5 < offsets[1] / total < 0.9
  for n in ('backbone.transformers.3.blocks.0.attn.query.weight', 'backbone.image_encoder.s4.b1.conv1.conv.weight', 'head.heatmap_head.0.weight',
            'join.layers.0.linear1.weight', 'backbone.c5_conv.weight'):
    assert finishes_early(n), n
  for n in ('backbone.transformers.2.blocks.1.mlp.0.weight', 'backbone.image_encoder.s3.b1.conv1.conv.weight', 'backbone.lidar_encoder.stem.conv.weight',
            'backbone.lidar_channel_to_img.2.weight'):
    assert not finishes_early(n), n
  # an observed assignment: four buckets in some completion order
  assign = {n: (i * 7) % 4 for i, n in enumerate(want)}
  model_cpu.__dict__['_grad_buckets'] = assign
  try:
    layout, total, offsets = arena_layout(model_cpu)
    assert len(offsets) == 5
    check(layout, total, offsets, lambda n: assign[n])
  finally:
    model_cpu.__dict__.pop('_grad_buckets', None)


@pytest.mark.gpu
def test_eval_forward_fp32_vs_reference_golden_and_oracle():
  """BASELINE config 2: TransFuser++ inference forward, bs=1, fp32, parity within 1e-3 relative."""
  m = _model().eval()
  inp = P.make_inputs(1)
  with torch.inference_mode():
    out = m(*[x.cuda() for x in inp])
    want = P.forward(P.make_state_dict(), P.PortConfig(), *inp)
  got = U.pack_outputs(out)
  errs = {}
  try:
    errs.update({'golden.' + k: v for k, v in U.compare_packed(got, U.load_golden('tfpp_eval_bs1.npz'), tol=1e9).items()})
    errs.update({'oracle.' + k: v for k, v in U.compare_packed(got, U.pack_outputs(want), tol=1e9).items()})
  finally:
    _report('eval_fp32', errs)
  bad = {k: v for k, v in errs.items() if v > U.REL_TOL_FP32}
  assert not bad, bad
  # the metric above is max|d| / max|ref| (one global scale, blind to errors on small elements): the three north-star outputs --
  # waypoints/checkpoints, target-speed logits, BEV heat-map -- are additionally held to |d| <= atol + rtol |ref| element-wise
  gold = U.load_golden('tfpp_eval_bs1.npz')
  for key, a in (('pred_checkpoint', out[2]), ('pred_target_speed', out[1]), ('bb_heatmap', out[6][0])):
    np.testing.assert_allclose(U.to_np(a), gold[key], rtol=1e-3, atol=1e-5 * float(np.abs(gold[key]).max()), err_msg=key)
  assert out[0] is None and out[7] is None and out[8] is None and out[9] is None and out[6][5] is None
  assert out[3].shape == (1, 7, 256, 1024) and out[5].shape == (1, 256, 1024) and out[3].dtype == torch.float32


@pytest.mark.gpu
def test_eval_forward_fp32_every_pixel_vs_oracle():
  """the golden comparison samples the dense maps (every 8th / 4th pixel + row sums).  Here EVERY element of every output --
  1.8 M semantic logits, 0.7 M BEV logits, 0.26 M depths, the five CenterNet maps, the planning outputs -- against the CPU oracle's full-resolution
  forward on the same inputs, for two different input seeds; tests/test_oracle.py::test_port_vs_live_reference holds the oracle to the unmodified
  reference at every element (2e-5), so a single wrong pixel anywhere on the HIP path fails."""
  m = _model().eval()
  worst = {}
  for seed in (1234, 7):
    inp = P.make_inputs(1, seed=seed)
    with torch.inference_mode():
      out = m(*[x.cuda() for x in inp])
      want = P.forward(P.make_state_dict(), P.PortConfig(), *inp)
    for k, e in U.assert_every_element_close(out, want, U.REL_TOL_FP32, f'HIP vs oracle (seed {seed})').items():
      worst[k] = max(worst.get(k, 0.0), e)
  _report('eval_fp32_every_pixel', worst)


@pytest.mark.gpu
def test_eval_forward_bf16_close_to_fp32_reference():
  """bf16 storage (training precision): the reference never validated reduced precision (config.py:245-246); the
  tolerance here is 5e-2 relative on the same golden vectors."""
  m = _model('bf16').eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1)])
  errs = U.compare_packed(U.pack_outputs(out), U.load_golden('tfpp_eval_bs1.npz'), tol=1e9)
  _report('eval_bf16', errs)
  bad = {k: v for k, v in errs.items() if v > 5e-2}
  assert not bad, bad


@pytest.mark.gpu
def test_aim_backbone_forward_and_train_step_vs_reference_golden():
  """BASELINE config 1 on the HIP path: the image-only AIM backbone (one RegNetY branch -> change_channel -> 257-token planning
  decoder).  Eval forward at bs = 1 within 1e-3 of the reference, train-mode step at bs = 2: losses, gradient norms and sampled
  gradient elements (tests/golden/tfpp_aim.npz, written by the reference through oracle/make_golden.py)."""
  from carla_garage_amd.engine import Tape
  from carla_garage_amd.losses import fused_losses, normalized_loss_weights
  g = U.load_golden('tfpp_aim.npz')
  m = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32', **AIM_CFG))
  full = P.make_state_dict()
  m.load_state_dict({k: full[k] for k in m.state_dict().keys()}, strict=True)
  m.cuda().eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1)])
  assert out[3] is None and out[4] is None and out[5] is None and out[6] is None
  e1 = U.assert_close(U.to_np(out[1]), g['eval_pred_target_speed'], U.REL_TOL_FP32, 'pred_target_speed')
  e2 = U.assert_close(U.to_np(out[2]), g['eval_pred_checkpoint'], U.REL_TOL_FP32, 'pred_checkpoint')
  m.train()
  _zero_dropout(m)
  eng = m._engine()
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  eng.prepare(torch.float32, True, True)
  eng.alloc_grads()
  eng.tape = Tape()
  t = eng.forward(*[x.cuda() for x in P.make_inputs(2)])
  w = normalized_loss_weights(m.config)
  assert w == {'loss_target_speed': 0.5, 'loss_checkpoint': 0.5}
  names, vals, seeds = fused_losses(m, t, batch, w, True)
  tape, eng.tape = eng.tape, None
  tape.backward(seeds)
  torch.cuda.synchronize()
  gl = dict(zip([str(x) for x in g['loss_names']], g['losses']))
  lerr = {n: float(abs(v - gl[n]) / abs(gl[n])) for n, v in zip(names, vals.cpu().numpy())}
  worst_norm, worst_elem = _compare_grads(eng, g)
  _report('aim', {'eval': [e1, e2], 'losses': lerr, 'grad_norm_worst': max(worst_norm.values()), 'grad_elem_worst': max(worst_elem.values()),
                  'tensors': len(worst_norm)})
  assert max(lerr.values()) <= 1e-3, lerr
  over = {n: e for n, e in worst_norm.items() if e > (GRAD_NORM_TOL_SE_FC1 if '.se.fc1.' in n else GRAD_NORM_TOL)}
  assert not over, over
  assert max(worst_elem.values()) <= GRAD_ELEM_TOL
  assert len(worst_norm) > 350  # every trainable tensor of the image branch and the planning head was compared


@pytest.mark.gpu
def test_wp_variant_forward():
  cfgw = dataclasses.replace(P.PortConfig(), use_wp_gru=True, use_controller_input_prediction=False)
  m = LidarCenterNet(GlobalConfig(use_wp_gru=True, use_controller_input_prediction=False))
  m.load_state_dict(P.make_state_dict(cfgw), strict=True)
  m.cuda().eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1, cfgw)])
  g = U.load_golden('tfpp_wp_eval_bs1.npz')
  U.assert_close(U.to_np(out[0]), g['pred_wp'], U.REL_TOL_FP32, 'pred_wp')
  U.assert_close(U.to_np(out[6][0]), g['bb_heatmap'], U.REL_TOL_FP32, 'heatmap')
  assert out[1] is None and out[2] is None


def _bev_model(dtype='fp32'):
  m = LidarCenterNet(GlobalConfig(backbone='bev_encoder', tfpp_dtype=dtype))
  m.load_state_dict(P.generic_state_dict(m.state_dict(), base=P.make_state_dict(P.PortConfig())), strict=True)
  return m.cuda()


@pytest.mark.gpu
def test_bev_encoder_eval_forward_fp32_vs_reference_golden():
  """backbone = 'bev_encoder' (team_code/bev_encoder.py) through the HIP path (carla_garage_amd/bev.py) against the unmodified reference:
  the depth layer, the camera -> BEV lift, the compressed BEV features, then every model output, 1e-3 relative."""
  g = U.load_golden('tfpp_bev_eval_bs1.npz')
  m = _bev_model().eval()
  assert list(m.state_dict().keys()) == [str(k) for k in g['keys']]
  eng = m._engine()
  eng.bev_runner.taps = {}
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1)])
  t = eng.bev_runner.taps
  errs = {'bev_depth_layer': U.rel_err(U.to_np(t['depth'].float().permute(0, 3, 1, 2))[:, ::4, ::4, ::8], g['bev_depth_layer']),
          'bev_lifted': U.rel_err(U.to_np(t['lifted'].float().permute(0, 3, 1, 2))[:, ::4, ::8, ::8], g['bev_lifted']),
          'bev_compressed': U.rel_err(U.to_np(t['compressed'].float().permute(0, 3, 1, 2))[:, ::4, ::8, ::8], g['bev_compressed'])}
  errs['pred_target_speed'] = U.rel_err(U.to_np(out[1]), g['pred_target_speed'])
  errs['pred_checkpoint'] = U.rel_err(U.to_np(out[2]), g['pred_checkpoint'])
  for i, n in enumerate(('heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res')):
    errs['bb_' + n] = U.rel_err(U.to_np(out[6][i]), g['bb_' + n])
  errs['pred_bev_semantic'] = U.rel_err(U.to_np(out[4])[:, :, ::U.BEV_STRIDE, ::U.BEV_STRIDE], g['pred_bev_semantic_strided'])
  errs['pred_semantic'] = U.rel_err(U.to_np(out[3])[:, :, ::U.SEM_STRIDE, ::U.SEM_STRIDE], g['pred_semantic_strided'])
  errs['pred_depth'] = U.rel_err(U.to_np(out[5])[:, ::U.DEPTH_STRIDE, ::U.DEPTH_STRIDE], g['pred_depth_strided'])
  _report('bev_eval_fp32_bs1', errs)
  assert max(errs.values()) <= 1e-3, errs


@pytest.mark.gpu
def test_bev_encoder_train_step_fp32_vs_reference_golden():
  """One train step at bs = 2 (batch-statistic BN in both RegNets, InstanceNorm, the lift's adjoint) against the unmodified reference.
  This configuration is ill-conditioned in fp32: the reference's OWN gradients in fp32 and fp64 (same weights, same batch, CPU) differ by
  4.9e-2 on image_encoder.s1.b1.se.fc1, 2.0e-2 on bev_encoder.s1.b2.se.fc1, ~1e-2 across both RegNets, and its losses by 3e-4 (yaw_res) /
  1e-4 (checkpoint) -- the BEV RegNet normalises 2 x 16 x 16 samples per channel in stage 3.  The bars are twice that spread; the new
  operators themselves are pinned tightly in tests/test_ops_gpu.py::test_bev_lift_and_instance_norm_vs_torch."""
  _check_train_step_vs_golden(2, 'tfpp_bev_train_bs2.npz', 'train_fp32_bev', model=_bev_model(), norm_tol=5e-2, norm_tol_se=1e-1, elem_tol=0.4)


def _swin_model(dtype):
  """BASELINE config 5: TransFuser++ with the Video-Swin LiDAR branch (6 LiDAR frames), deterministic weights by name."""
  cfg = GlobalConfig(lidar_architecture='video_swin_tiny', lidar_seq_len=6, tfpp_dtype=dtype)
  m = LidarCenterNet(cfg)
  m.load_state_dict(P.generic_state_dict(m.state_dict(), base=P.make_state_dict(P.PortConfig())), strict=True)
  return m.cuda().eval(), dataclasses.replace(P.PortConfig(), lidar_seq_len=6)


@pytest.mark.gpu
def test_video_swin_eval_forward_fp32_vs_reference_golden():
  """team_code/video_swin_transformer.py through the HIP path (carla_garage_amd/swin.py) against the unmodified reference
  (tests/golden/tfpp_swin_eval_bs1.npz, oracle/make_golden.py swin): the taps of the LiDAR branch, then the model outputs incl. the
  temporal velocity / brake heads, 1e-3 relative (north_star)."""
  g = U.load_golden('tfpp_swin_eval_bs1.npz')
  m, pc = _swin_model('fp32')
  assert list(m.state_dict().keys()) == [str(k) for k in g['keys']]
  eng = m._engine()
  eng.swin.taps = {}
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1, pc)])
  errs = {}
  for name, t in eng.swin.taps.items():  # (B, D, H, W, C) -> the golden's (B, C, D, H, W) strided view
    got = U.to_np(t.float().permute(0, 4, 1, 2, 3))[:, ::8, :, ::4, ::4]
    errs[name] = U.rel_err(got, g[name])
  errs['pred_target_speed'] = U.rel_err(U.to_np(out[1]), g['pred_target_speed'])
  errs['pred_checkpoint'] = U.rel_err(U.to_np(out[2]), g['pred_checkpoint'])
  for i, n in enumerate(('heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res', 'velocity', 'brake')):
    errs['bb_' + n] = U.rel_err(U.to_np(out[6][i]), g['bb_' + n])
  errs['pred_bev_semantic'] = U.rel_err(U.to_np(out[4])[:, :, ::U.BEV_STRIDE, ::U.BEV_STRIDE], g['pred_bev_semantic_strided'])
  errs['pred_semantic'] = U.rel_err(U.to_np(out[3])[:, :, ::U.SEM_STRIDE, ::U.SEM_STRIDE], g['pred_semantic_strided'])
  _report('swin_eval_fp32_bs1', errs)
  assert max(errs.values()) <= 1e-3, errs


@pytest.mark.gpu
def test_video_swin_train_step_fp32_vs_reference_golden():
  """BASELINE config 5 trains: one step at bs = 2 (stochastic depth and dropout off on both sides) against the unmodified reference --
  the 12 losses, gradient norms and sampled gradient elements of every parameter (Swin blocks, relative-position bias tables, patch
  embedding / merging, Conv3d adapters), BN running statistics of the image branch."""
  m = LidarCenterNet(GlobalConfig(lidar_architecture='video_swin_tiny', lidar_seq_len=6))
  m.load_state_dict(P.generic_state_dict(m.state_dict(), base=P.make_state_dict(P.PortConfig())), strict=True)
  m.cuda()
  m._engine().swin.drop_path_rate = 0.0
  _check_train_step_vs_golden(2, 'tfpp_swin_train_bs2.npz', 'train_fp32_swin', model=m, port_cfg=dataclasses.replace(P.PortConfig(), lidar_seq_len=6))


@pytest.mark.gpu
def test_video_swin_eval_forward_bf16_close_to_fp32_reference():
  g = U.load_golden('tfpp_swin_eval_bs1.npz')
  m, pc = _swin_model('bf16')
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1, pc)])
  errs = {'pred_target_speed': U.rel_err(U.to_np(out[1]), g['pred_target_speed']), 'pred_checkpoint': U.rel_err(U.to_np(out[2]), g['pred_checkpoint']),
          'bb_heatmap': U.rel_err(U.to_np(out[6][0]), g['bb_heatmap'])}
  _report('swin_eval_bf16_bs1', errs)
  assert max(errs.values()) <= 5e-2, errs


def _zero_dropout(m):
  for mod in m.modules():
    if isinstance(mod, torch.nn.Dropout):
      mod.p = 0.0
  m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0


def _engine_train_step(m, bs, port_cfg=None):
  """train-mode forward (batch statistics, dropout 0), fused losses, hand-written backward on the engine; returns
  (loss names, loss values [numpy], engine with .grads filled)."""
  from carla_garage_amd.engine import Tape
  from carla_garage_amd.losses import fused_losses, normalized_loss_weights
  _zero_dropout(m)
  eng = m._engine()
  batch = {k: v.cuda() for k, v in P.make_labels(bs, port_cfg).items()}
  inp = [x.cuda() for x in P.make_inputs(bs, port_cfg)]
  eng.prepare(m.compute_dtype, True, True)
  eng.alloc_grads()
  eng.tape = Tape()
  t = eng.forward(*inp)
  names, vals, seeds = fused_losses(m, t, batch, normalized_loss_weights(m.config), True)
  tape, eng.tape = eng.tape, None
  tape.backward(seeds)
  torch.cuda.synchronize()
  return names, vals.cpu().numpy(), eng


# Gradient parity bars.  Two CPU fp32 implementations of this network (the reference and oracle/tfpp_port.py) differ by up to
# 5.4e-3 in per-tensor gradient norms and by up to 0.08 x (rms + |ref|) on single elements (bs = 12, LiDAR-branch BatchNorm biases:
# train-mode BN makes the gradients ill-conditioned), so: norms at 1e-2, sampled ELEMENTS at |d| <= 0.15 x (rms + |ref|) -- a
# transposed or permuted weight gradient misses that by an order of magnitude, which a norm check cannot see.
GRAD_NORM_TOL, GRAD_ELEM_TOL = 1e-2, 0.15
# ... except the squeeze-excite reduction layers (se.fc1: 8..144 outputs fed by a global mean, the most ill-conditioned gradients of
# the network): measured 1.09e-2 / 1.01e-2 on lidar_encoder.s2.b1.se.fc1 at bs = 12 (everything else <= 7e-3), bar 2e-2
GRAD_NORM_TOL_SE_FC1 = 2e-2


def _compare_grads(eng, g):
  worst_norm, worst_elem = {}, {}
  for name, (norm, gmax), samples in zip(g['grad_names'], g['grad_norms'], g['grad_samples']):
    if gmax < 1e-5:  # structurally zero (key biases, pre-BN biases)
      continue
    mine = eng.grads[str(name)].detach().flatten()
    worst_norm[str(name)] = abs(mine.double().norm().item() - norm) / norm
    idx = U.sample_idx(mine.numel())
    got = mine[torch.from_numpy(idx).to(mine.device)].float().cpu().numpy()
    ref = samples[:len(idx)]
    rms = norm / np.sqrt(mine.numel())
    worst_elem[str(name)] = float(np.max(np.abs(got - ref) / (rms + np.abs(ref))))
  return worst_norm, worst_elem


def _check_train_step_vs_golden(bs, fname, tag, model=None, port_cfg=None, norm_tol=None, norm_tol_se=None, elem_tol=None, norm_tol_where=None):
  g = U.load_golden(fname)
  m = (model if model is not None else _model()).train()
  names, vals, eng = _engine_train_step(m, bs, port_cfg)
  gl = dict(zip([str(x) for x in g['loss_names']], g['losses']))
  errs = {n: float(abs(v - gl[n]) / abs(gl[n])) for n, v in zip(names, vals)}
  worst_norm, worst_elem = _compare_grads(eng, g)
  top = dict(sorted(worst_norm.items(), key=lambda kv: -kv[1])[:8])
  tope = dict(sorted(worst_elem.items(), key=lambda kv: -kv[1])[:8])
  sd = m.state_dict()
  rs = {str(n): abs(float(sd[str(n)].double().sum()) - s) / (abs(s) + 1.0) for n, s in zip(g['running_names'], g['running_sums'])}
  _report(tag, {'losses': errs, 'grad_norm_worst': top, 'grad_elem_worst': tope, 'running_worst': max(rs.values())})
  assert max(errs.values()) <= 1e-3, errs
  tol, tol_se = norm_tol or GRAD_NORM_TOL, norm_tol_se or GRAD_NORM_TOL_SE_FC1
  where = norm_tol_where or {}  # {substring of the parameter name: bar} for tensors a test documents as ill-conditioned
  bar = lambda n: next((t for sub, t in where.items() if sub in n), tol_se if '.se.fc' in n else tol)
  over = {n: e for n, e in worst_norm.items() if e > bar(n)}
  assert not over, over
  assert max(worst_elem.values()) <= (elem_tol or GRAD_ELEM_TOL), tope
  assert max(rs.values()) <= 1e-3
  return m, eng


@pytest.mark.gpu
def test_train_step_fp32_vs_reference_golden():
  """train-mode forward (batch statistics), fused losses, hand-written backward at bs = 2: losses 1e-3, per-parameter gradient
  norms and sampled gradient elements (see GRAD_*_TOL), BN running statistics."""
  _check_train_step_vs_golden(2, 'tfpp_train_bs2.npz', 'train_fp32')


@pytest.mark.gpu
def test_train_step_bs12_fp32_vs_reference_golden():
  """BASELINE config 3's batch size.  bs = 12 is where the dispatcher switches to the kernels bench.py runs (8-wave 128x128 LDS-DMA
  tiles with fused BN statistics and the M-major XCD order, the bs=12 weight-gradient plans): the reference's own step at that size
  (tests/golden/tfpp_train_bs12.npz, oracle/make_golden.py) is compared with the fp32 HIP step."""
  _check_train_step_vs_golden(12, 'tfpp_train_bs12.npz', 'train_fp32_bs12')


@pytest.mark.gpu
def test_focal_target_speed_loss_train_step_fp32_vs_reference_golden():
  """config.use_focal_loss = 1 (team_code/model.py:255-256, focal_loss.py): the ten losses, every gradient and the state_dict schema of the loss
  container (loss_speed.nll_loss.weight) against one train step of the unmodified reference (oracle/make_golden.py focal)."""
  g = U.load_golden('tfpp_train_focal_bs2.npz')
  m = LidarCenterNet(GlobalConfig(use_focal_loss=True))
  assert [k for k in m.state_dict().keys() if k.startswith('loss_speed')] == [str(k) for k in g['state_dict_keys_loss_speed']] == ['loss_speed.nll_loss.weight']
  sd = P.make_state_dict(P.PortConfig())
  sd['loss_speed.nll_loss.weight'] = sd.pop('loss_speed.weight')
  m.load_state_dict(sd, strict=True)
  plain = U.load_golden('tfpp_train_bs2.npz')
  lp, lf = dict(zip(map(str, plain['loss_names']), plain['losses'])), dict(zip(map(str, g['loss_names']), g['losses']))
  assert abs(lp['loss_target_speed'] - lf['loss_target_speed']) > 0.1 * lp['loss_target_speed']   # it IS another loss than the cross entropy
  _check_train_step_vs_golden(2, 'tfpp_train_focal_bs2.npz', 'train_fp32_focal', model=m.cuda())


@pytest.mark.gpu
def test_temporal_lidar_train_step_fp32_vs_reference_golden():
  """lidar_seq_len = 6 on the default RegNet LiDAR branch (6 BEV frames as input channels): the velocity / brake CenterNet heads and
  their losses (center_net.py:29-31,119-123) -- losses, gradients and BN statistics of one train step against the unmodified reference."""
  m = LidarCenterNet(GlobalConfig(lidar_seq_len=6))
  m.load_state_dict(P.generic_state_dict(m.state_dict(), base=P.make_state_dict(P.PortConfig())), strict=True)
  g = U.load_golden('tfpp_train_temporal_bs2.npz')
  assert [str(x) for x in g['loss_names']][-2:] == ['loss_velocity', 'loss_brake']
  _check_train_step_vs_golden(2, 'tfpp_train_temporal_bs2.npz', 'train_fp32_temporal', model=m.cuda(),
                              port_cfg=dataclasses.replace(P.PortConfig(), lidar_seq_len=6))


@pytest.mark.gpu
def test_wp_variant_train_step_fp32_vs_reference_golden():
  """The waypoint variant trained (VERDICT r2 missing #5): wp_query, the 8-step GRU decoder and loss_wp (model.py:165-171,333-334,399-404) --
  losses, per-parameter gradient norms / sampled elements and BN statistics of one step against the unmodified reference
  (tests/golden/tfpp_wp_train_bs2.npz, `python -m oracle.make_golden wp_train`)."""
  cfgw = dataclasses.replace(P.PortConfig(), use_wp_gru=True, use_controller_input_prediction=False)
  m = LidarCenterNet(GlobalConfig(use_wp_gru=True, use_controller_input_prediction=False))
  m.load_state_dict(P.make_state_dict(cfgw), strict=True)
  g = U.load_golden('tfpp_wp_train_bs2.npz')
  assert 'loss_wp' in [str(x) for x in g['loss_names']] and 'wp_decoder.gru.weight_hh_l0' in [str(x) for x in g['grad_names']]
  _check_train_step_vs_golden(2, 'tfpp_wp_train_bs2.npz', 'train_fp32_wp', model=m.cuda(), port_cfg=cfgw)


MULTI_WP = dict(use_wp_gru=True, use_controller_input_prediction=False, multi_wp_output=True)


def _multi_wp_model(dtype='fp32'):
  from oracle.make_golden import MULTI_WP_LABEL_SEED
  cfgm = dataclasses.replace(P.PortConfig(), extra={'label_seed': MULTI_WP_LABEL_SEED}, **MULTI_WP)
  m = LidarCenterNet(GlobalConfig(tfpp_dtype=dtype, **MULTI_WP))
  m.load_state_dict(P.make_state_dict(cfgm), strict=True)
  return m.cuda(), cfgm


def test_multi_wp_variant_schema_and_loss_weights():
  """CPU: the multi_wp_output module has the reference's state_dict (keys, order, shapes: strict load) and loss list (loss_selection right behind
  loss_wp, both at the same normalised weight, train.py:440-441)."""
  from carla_garage_amd.losses import active_losses, normalized_loss_weights, output_slots
  from oracle.make_golden import MULTI_WP_LABEL_SEED
  cfgm = dataclasses.replace(P.PortConfig(), extra={'label_seed': MULTI_WP_LABEL_SEED}, **MULTI_WP)
  m = LidarCenterNet(GlobalConfig(**MULTI_WP))
  g = U.load_golden('tfpp_multi_wp_eval_bs1.npz')
  assert list(m.state_dict().keys()) == [str(k) for k in g['state_dict_keys']]
  m.load_state_dict(P.make_state_dict(cfgm), strict=True)
  assert 0.0 <= float(m.wp_query.min()) and tuple(m.wp_query.shape) == (1, 17, 256)
  names = active_losses(m.config)
  assert names[:2] == ['loss_wp', 'loss_selection'] and names == [str(x) for x in U.load_golden('tfpp_multi_wp_train_bs4.npz')['loss_names']]
  assert [k for k, _ in output_slots(m.config)][:3] == ['loss_wp', 'loss_wp/1', 'loss_selection']
  w = normalized_loss_weights(m.config)
  assert w['loss_selection'] == w['loss_wp'] and abs(sum(w.values()) - 1.0) < 1e-12
  # ... and the default configuration is untouched by the variant's code
  assert 'loss_selection' not in active_losses(GlobalConfig()) and all(k == l for k, l in output_slots(GlobalConfig()))


@pytest.mark.gpu
def test_multi_wp_variant_forward():
  """config.multi_wp_output (config.py:484; model.py:151-163,326-331): wp_query (1, 17, 256), two GRU decoders and the select_wps logit -- the
  reference's state_dict loads strictly and pred_wp / pred_wp_1 / selected_path (tuple slots 0, 8, 9) agree with the unmodified reference."""
  m, cfgm = _multi_wp_model()
  g = U.load_golden('tfpp_multi_wp_eval_bs1.npz')
  assert list(m.state_dict().keys()) == [str(k) for k in g['state_dict_keys']]
  m.eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1, cfgm)])
  assert tuple(out[0].shape) == (1, 8, 2) and tuple(out[8].shape) == (1, 8, 2) and tuple(out[9].shape) == (1, 1)
  for i, k in ((0, 'pred_wp'), (8, 'pred_wp_1'), (9, 'selected_path')):
    U.assert_close(U.to_np(out[i]), g[k], U.REL_TOL_FP32, k)
  U.assert_close(U.to_np(out[6][0]), g['bb_heatmap'], U.REL_TOL_FP32, 'heatmap')
  assert out[1] is None and out[2] is None and out[7] is None


@pytest.mark.gpu
def test_multi_wp_variant_train_step_fp32_vs_reference_golden():
  """loss_wp = mean_b min over the two hypotheses, loss_selection = BCE of the logit against the arg-min (model.py:401-411; weight 1.0:
  train.py:440-441): losses, per-parameter gradient norms / sampled elements and BN statistics of one step at bs = 4 against the unmodified
  reference -- two samples train hypothesis 0, two hypothesis 1 (tests/golden/tfpp_multi_wp_train_bs4.npz, `python -m oracle.make_golden multi_wp`)."""
  m, cfgm = _multi_wp_model()
  g = U.load_golden('tfpp_multi_wp_train_bs4.npz')
  assert sorted(g['selection_labels'].tolist()) == [0, 0, 1, 1]
  names = [str(x) for x in g['grad_names']]
  assert all(k in names for k in ('wp_decoder_1.gru.weight_hh_l0', 'select_wps.weight', 'select_wps.bias', 'wp_query'))
  _check_train_step_vs_golden(4, 'tfpp_multi_wp_train_bs4.npz', 'train_fp32_multi_wp', model=m, port_cfg=cfgm)


def _drive_like_train_py(make_model, port_cfg, bs, gname, tag, tamper=None, norm_tol_where=None):
  """forward -> compute_loss -> weighted sum -> backward as team_code/train.py:776-898 drives the module, five times on one module (eager steps, then the
  captured hipGraphs), every time against the losses and gradients the unmodified reference wrote; then once more on a fresh module with
  ``tamper(out)`` applied to the forward's tuple, which sends compute_loss down its general path (gradients in the caller layout through autograd)."""
  from carla_garage_amd.losses import normalized_loss_weights
  g = U.load_golden(gname)
  lab = {k: v.cuda() for k, v in P.make_labels(bs, port_cfg).items()}
  inp = [x.cuda() for x in P.make_inputs(bs, port_cfg)]

  def step(m, change=None):
    m.zero_grad(set_to_none=True)
    out = list(m(*inp))
    if change is not None:
      out = change(out)
    losses = m.compute_loss(pred_wp=out[0], pred_target_speed=out[1], pred_checkpoint=out[2], pred_semantic=out[3], pred_bev_semantic=out[4],
                            pred_depth=out[5], pred_bounding_box=out[6], pred_wp_1=out[8], selected_path=out[9], **lab)
    assert list(losses.keys()) == [str(x) for x in g['loss_names']]
    w = normalized_loss_weights(m.config)
    total = sum(w[k] * v for k, v in losses.items())
    total.backward()
    np.testing.assert_allclose(np.array([float(v.detach()) for v in losses.values()]), g['losses'], rtol=1e-3)
    np.testing.assert_allclose(float(total.detach()), float(g['total_loss']), rtol=1e-3)
    return out

  m = make_model().train()
  _zero_dropout(m)
  modes, outs = [], []
  for i in range(5):
    outs.append(step(m))
    modes.append(m._dropin().cur['mode'])
    _param_grads_vs_golden(m, g, f'{tag}_{i}', norm_tol_where)
  assert modes[0] == 'eager' and modes[-1] == 'graph', modes
  if tamper is not None:
    m2 = make_model().train()  # (a fresh module: the first one replays its captured graphs by now, and those only take token gradients)
    _zero_dropout(m2)
    step(m2, tamper)
    assert m2._dropin().cur['mode'] == 'eager'
    _param_grads_vs_golden(m2, g, f'{tag}_general', norm_tol_where)
  return m, outs


@pytest.mark.gpu
def test_multi_wp_variant_through_the_reference_call_sequence():
  """compute_loss(pred_wp, pred_wp_1, selected_path, ...) on the token path (loss_wp hands ONE token gradient to both hypotheses), eager and
  replayed, and on the general path (pred_wp_1 replaced by a copy of itself)."""
  from carla_garage_amd.losses import normalized_loss_weights
  m, cfgm = _multi_wp_model()
  w = normalized_loss_weights(m.config)
  assert abs(w['loss_selection'] - w['loss_wp']) < 1e-12 and abs(sum(w.values()) - 1.0) < 1e-9

  def copy_second(out):
    out[8] = out[8] * 1.0
    return out

  _drive_like_train_py(lambda: _multi_wp_model()[0], cfgm, 4, 'tfpp_multi_wp_train_bs4.npz', 'dropin_multi_wp', tamper=copy_second)


def _tp_attention_model(dtype='fp32'):
  m = LidarCenterNet(GlobalConfig(tfpp_dtype=dtype, tp_attention=True))
  m.load_state_dict(P.generic_state_dict(m.state_dict(), base=P.make_state_dict(P.PortConfig())), strict=True)
  return m.cuda(), dataclasses.replace(P.PortConfig(), tp_attention=True)


def test_tp_attention_variant_schema():
  """CPU: the tp_attention module has the reference's state_dict (tp_pos_embed before extra_sensor_pos_embed, tp_encoder, the decoder's separate
  key / query / value / proj linears) and refuses the combination the reference's forward cannot run (with use_wp_gru: model.py:332-334)."""
  g = U.load_golden('tfpp_tp_attention_eval_bs1.npz')
  m = LidarCenterNet(GlobalConfig(tp_attention=True))
  assert list(m.state_dict().keys()) == [str(k) for k in g['state_dict_keys']]
  assert isinstance(m.join.layers[0].activation, torch.nn.GELU) and 0.0 <= float(m.tp_pos_embed.min())
  with pytest.raises(ValueError):
    LidarCenterNet(GlobalConfig(tp_attention=True, use_wp_gru=True))


@pytest.mark.gpu
def test_tp_attention_variant_forward():
  """config.tp_attention (config.py:483; model.py:124-134,336-350; transfuser.py:404-508): predictions within 1e-3 of the unmodified reference and the
  [vision, speed, target point] attention read-out in tuple slot 7 (three host floats that sum to one)."""
  m, cfgt = _tp_attention_model()
  g = U.load_golden('tfpp_tp_attention_eval_bs1.npz')
  m.eval()
  with torch.inference_mode():
    out = m(*[x.cuda() for x in P.make_inputs(1, cfgt)])
  U.assert_close(U.to_np(out[1]), g['pred_target_speed'], U.REL_TOL_FP32, 'pred_target_speed')
  U.assert_close(U.to_np(out[2]), g['pred_checkpoint'], U.REL_TOL_FP32, 'pred_checkpoint')
  U.assert_close(U.to_np(out[6][0]), g['bb_heatmap'], U.REL_TOL_FP32, 'heatmap')
  assert isinstance(out[7], list) and len(out[7]) == 3 and all(isinstance(v, float) for v in out[7])
  np.testing.assert_allclose(np.array(out[7]), g['attention_weights'], rtol=1e-3)
  assert abs(sum(out[7]) - 1.0) < 1e-5
  _report('tp_attention_eval', {'attention_weights': out[7], 'reference': g['attention_weights'].tolist()})
  # the default module keeps returning None there
  d = _model().eval()
  with torch.inference_mode():
    assert d(*[x.cuda() for x in P.make_inputs(1)])[7] is None


@pytest.mark.gpu
def test_tp_attention_variant_train_step_fp32_vs_reference_golden():
  """One train step at bs = 2: losses, per-parameter gradient norms / sampled elements (the decoder's q / k / v / proj linears, tp_encoder, tp_pos_embed
  among them) and BN statistics against the unmodified reference (`python -m oracle.make_golden tp_attention`)."""
  m, cfgt = _tp_attention_model()
  g = U.load_golden('tfpp_tp_attention_train_bs2.npz')
  names = [str(x) for x in g['grad_names']]
  assert all(k in names for k in ('tp_pos_embed', 'tp_encoder.0.weight', 'tp_encoder.2.bias', 'join.layers.0.self_attn.query.weight',
                                  'join.layers.5.multihead_attn.key.weight', 'join.layers.3.multihead_attn.proj.bias'))
  # The first decoder layer's self-attention sees the query parameter itself (the same rows for every sample); the gradients of its query / key
  # linears are ill-conditioned in fp32: the REFERENCE's own fp32 norms are 4.3e-3 off a float64 evaluation of the same step (the CPU port in double
  # precision, stored in the fixture as grad_norms_fp64), the port's fp32 norms 2.9e-3.  Those tensors are held to 2e-2 against the reference's fp32
  # values and to 1e-2 against the float64 ones; every other tensor keeps the bars of the default step.
  ill = 'join.layers.0.self_attn.'
  _, eng = _check_train_step_vs_golden(2, 'tfpp_tp_attention_train_bs2.npz', 'train_fp32_tp_attention', model=m, port_cfg=cfgt,
                                       norm_tol_where={ill + 'query': 2e-2, ill + 'key': 2e-2})
  vs64 = {}
  for name, (_, gmax), n64 in zip(names, g['grad_norms'], g['grad_norms_fp64']):
    if gmax >= 1e-5 and name.startswith('join.'):
      vs64[name] = abs(eng.grads[name].double().norm().item() - n64) / n64
  _report('train_fp32_tp_attention_vs_fp64', dict(sorted(vs64.items(), key=lambda kv: -kv[1])[:6]))
  assert max(vs64.values()) <= 1e-2, dict(sorted(vs64.items(), key=lambda kv: -kv[1])[:6])


@pytest.mark.gpu
def test_tp_attention_variant_through_the_reference_call_sequence():
  """The variant behind forward -> compute_loss -> backward, eager and replayed (the attention read-out of a replayed forward is read from the graph's
  own accumulator after the replay), and on compute_loss's general path."""

  def copy_checkpoints(out):
    out[2] = out[2] * 1.0
    return out

  g = U.load_golden('tfpp_tp_attention_train_bs2.npz')
  ill = 'join.layers.0.self_attn.'  # (ill-conditioned in fp32: see test_tp_attention_variant_train_step_fp32_vs_reference_golden)
  m, outs = _drive_like_train_py(lambda: _tp_attention_model()[0], _tp_attention_model()[1], 2, 'tfpp_tp_attention_train_bs2.npz', 'dropin_tp_attention',
                                 tamper=copy_checkpoints, norm_tol_where={ill + 'query': 2e-2, ill + 'key': 2e-2})
  for o in outs:  # same weights, same batch: the read-out of the replayed steps is that of the eager ones
    assert len(o[7]) == 3 and abs(sum(o[7]) - 1.0) < 1e-5
    np.testing.assert_allclose(np.array(o[7]), np.array(outs[0][7]), rtol=1e-5)


BF16_LOSS_TOL = 5e-2  # measured worst: loss_yaw_res 3.4e-2 (a loss of ~1e-2 absolute, i.e. 3e-4 absolute error), every other loss <= 5e-3
# Gradients of the bf16 step against the fp32 HIP step on identical weights and batch (tools/bf16_evidence.py, round 3; 568 tensors carrying
# >= 1e-3 of the largest norm).  bf16 rounding through 100+ layers with batch-statistic BatchNorm is chaotic on the ill-conditioned tensors
# (SE fc1, attention query / key: differences of nearly cancelling terms -- a change of the summation order inside one kernel moved
# lidar_encoder.s2.b1.se.fc1.weight from 0.33 to 0.75), so the bars are on robust statistics, each about 1.3x what is measured:
#   whole gradient arena: cosine 0.9940, relative L2 0.110;  per-tensor norm error: median 0.0083, p90 0.071, p99 0.226, max 0.745;
#   sampled elements (16 per tensor, 9068): 9.5 % beyond 0.5 x (rms + |ref|)
BF16_ARENA_COSINE, BF16_ARENA_REL_L2 = 0.99, 0.15
# Round 6 measured the TAIL of that distribution over eight builds that differ only in where one intermediate of the squeeze-excite block is
# rounded to bf16 (profiles/r06_bf16_tail_statistics.txt): p90 0.062-0.086, p99 0.17-0.33, max 0.48-1.31 -- the p99 / max bars of rounds 3-5
# (0.30 / 1.0, 1.3x one measurement) sat inside that spread.  Median and the arena statistics do not move.
BF16_NORM_MEDIAN, BF16_NORM_P90, BF16_NORM_P99, BF16_NORM_MAX = 0.012, 0.11, 0.42, 1.7
BF16_ELEM_FRACTION_BEYOND_HALF = 0.13


@pytest.mark.gpu
def test_train_step_bs12_bf16_vs_fp32_hip_and_golden():
  """The benchmarked precision: the bf16 step at bs = 12 against the reference's fp32 losses and against the fp32 HIP step's gradients --
  losses, the whole gradient arena (cosine / relative L2), the distribution of per-tensor norm errors and sampled ELEMENTS (see the
  constants above).  tests/test_model.py::test_bf16_trains_like_fp32_over_200_steps_... is the end-to-end counterpart."""
  g = U.load_golden('tfpp_train_bs12.npz')
  m32 = _model('fp32').train()
  _, v32, e32 = _engine_train_step(m32, 12)
  ref = {n: e32.grads[n].detach().clone() for n in e32.grads}
  del m32, e32
  torch.cuda.empty_cache()
  m16 = _model('bf16').train()
  names, v16, e16 = _engine_train_step(m16, 12)
  gl = dict(zip([str(x) for x in g['loss_names']], g['losses']))
  lerr = {n: float(abs(v - gl[n]) / abs(gl[n])) for n, v in zip(names, v16)}
  flat32 = torch.cat([ref[n].flatten().double() for n in ref])
  flat16 = torch.cat([e16.grads[n].detach().flatten().double() for n in ref])
  cosine = float((flat32 * flat16).sum() / (flat32.norm() * flat16.norm()))
  rel_l2 = float((flat16 - flat32).norm() / flat32.norm())
  big = max(float(r.double().norm()) for r in ref.values())
  nerr, beyond, total = {}, 0, 0
  for n, r in ref.items():
    rn = float(r.double().norm())
    if rn < 1e-3 * big:
      continue
    mine = e16.grads[n].detach()
    nerr[n] = abs(float(mine.double().norm()) - rn) / rn
    idx = torch.from_numpy(U.sample_idx(r.numel())).to(r.device)
    rs, gs = r.flatten()[idx].double(), mine.flatten()[idx].double()
    rel = (gs - rs).abs() / (rn / np.sqrt(r.numel()) + rs.abs())
    beyond += int((rel > 0.5).sum())
    total += int(rel.numel())
  e = np.array(list(nerr.values()))
  stats = {'arena_cosine': cosine, 'arena_rel_l2': rel_l2, 'norm_err_median': float(np.median(e)), 'norm_err_p90': float(np.percentile(e, 90)),
           'norm_err_p99': float(np.percentile(e, 99)), 'norm_err_max': float(e.max()), 'sampled_elements': total, 'beyond_half': beyond / total}
  _report('train_bf16_bs12', {'losses_vs_reference': lerr, 'losses_vs_fp32_hip': {n: float(abs(a - b) / abs(b)) for n, a, b in zip(names, v16, v32)},
                               'gradients_vs_fp32_hip': stats, 'tensors_compared': len(nerr),
                               'norm_err_top': dict(sorted(nerr.items(), key=lambda kv: -kv[1])[:6])})
  assert np.isfinite(v16).all()
  assert max(lerr.values()) <= BF16_LOSS_TOL, lerr
  assert cosine >= BF16_ARENA_COSINE and rel_l2 <= BF16_ARENA_REL_L2, stats
  assert stats['norm_err_median'] <= BF16_NORM_MEDIAN and stats['norm_err_p90'] <= BF16_NORM_P90 and stats['norm_err_p99'] <= BF16_NORM_P99 and \
      stats['norm_err_max'] <= BF16_NORM_MAX, stats
  assert stats['beyond_half'] <= BF16_ELEM_FRACTION_BEYOND_HALF, stats


@pytest.mark.gpu
def test_bf16_step_is_no_worse_than_the_autocast_reference():
  """Pins the benchmarked precision against a bf16 REFERENCE: tests/golden/tfpp_bf16_autocast_bs12.npz holds what the
  unmodified reference does to its own gradients when train.py:885's autocast runs in bfloat16 (oracle/make_golden_bf16.py: CPU autocast,
  deterministic test weights, the bs = 12 batch, dropout 0).  The HIP bf16 step -- against the HIP fp32 step on the same weights and batch,
  same formulas (oracle/grad_stats.py) -- must be no worse on any statistic, and its losses no further from the reference's fp32 losses
  than twice the autocast reference's own distance."""
  from oracle.grad_stats import STAT_KEYS, gradient_stats
  g = U.load_golden('tfpp_bf16_autocast_bs12.npz')
  want = dict(zip([str(k) for k in g['stat_names']], [float(v) for v in g['stats_autocast_vs_fp32']]))
  m32 = _model('fp32').train()
  _, v32, e32 = _engine_train_step(m32, 12)
  ref = {n: e32.grads[n].detach().clone() for n in e32.grads}
  del m32, e32
  torch.cuda.empty_cache()
  m16 = _model('bf16').train()
  names, v16, e16 = _engine_train_step(m16, 12)
  got = gradient_stats(ref, {n: e16.grads[n].detach() for n in ref})
  l_ref = dict(zip([str(x) for x in g['loss_names']], g['losses_fp32']))
  l_ac = dict(zip([str(x) for x in g['loss_names']], g['losses_autocast']))
  l_hip = dict(zip(names, v16))
  loss_dev = {n: (abs(float(l_hip[n]) - l_ref[n]) / abs(l_ref[n]), abs(l_ac[n] - l_ref[n]) / abs(l_ref[n])) for n in names}
  # the HIP bf16 gradients against the autocast reference's gradients directly: two bf16 roundings of the same fp32 quantity
  an = dict(zip([str(x) for x in g['grad_names']], g['autocast_grad_norms']))
  fn = dict(zip([str(x) for x in g['grad_names']], g['fp32_grad_norms']))
  big = max(fn.values())
  direct = sorted(abs(float(e16.grads[n].double().norm()) - an[n]) / an[n] for n in ref if n in an and fn[n] >= 1e-3 * big)
  _report('bf16_vs_autocast_reference', {'hip_bf16_vs_hip_fp32': got, 'reference_autocast_vs_reference_fp32': want,
                                         'losses_rel_dev_hip_and_autocast': {n: [float(a), float(b)] for n, (a, b) in loss_dev.items()},
                                         'hip_bf16_norms_vs_autocast_norms': {'median': direct[len(direct) // 2], 'p90': direct[int(0.9 * len(direct))], 'max': direct[-1]}})
  # the arena statistics, the median and the element count are stable to the last digit between builds; p90 / p99 / max of the per-tensor norm
  # error are set by a handful of squeeze-excite fc1 tensors and move by up to 2x between arithmetically equivalent builds (round 6: eight
  # variants, profiles/r06_bf16_tail_statistics.txt: p99 0.17-0.33 against the reference's single draw of 0.297): those three get that spread
  slack = 1.02
  tail = {'norm_err_p90': 1.3, 'norm_err_p99': 1.4, 'norm_err_max': 1.5}
  assert got['arena_cosine'] >= 1.0 - (1.0 - want['arena_cosine']) * slack, (got, want)
  for k in STAT_KEYS[1:]:
    assert got[k] <= want[k] * tail.get(k, slack), (k, got, want)
  assert got['tensors'] >= 500
  for n, (a, b) in loss_dev.items():
    assert a <= 2.0 * b + 1e-3, (n, a, b)


@pytest.mark.gpu
def test_video_swin_bf16_step_at_the_benchmarked_batch_is_no_worse_than_the_autocast_reference():
  """BASELINE config 5 at the setting bench.py times (Video-Swin LiDAR branch, 6 LiDAR frames, bs = 4, bf16):
  tests/golden/tfpp_swin_bf16_autocast_bs4.npz holds what bfloat16 autocast does to the unmodified reference's own gradients on this
  configuration (oracle/make_golden_bf16.py swin).  The HIP bf16 step -- against the HIP fp32 step, same weights, same batch, same statistics
  (oracle/grad_stats.py) -- must be no worse; the tail statistics get the spread measured on the default configuration
  (profiles/r06_bf16_tail_statistics.txt), the losses twice the autocast reference's own distance from fp32."""
  from oracle.grad_stats import STAT_KEYS, gradient_stats
  g = U.load_golden('tfpp_swin_bf16_autocast_bs4.npz')
  want = dict(zip([str(k) for k in g['stat_names']], [float(v) for v in g['stats_autocast_vs_fp32']]))
  bs = int(g['batch'])

  def step(dtype):
    m, pc = _swin_model(dtype)
    m.train()
    m._engine().swin.drop_path_rate = 0.0
    return _engine_train_step(m, bs, pc)

  _, v32, e32 = step('fp32')
  ref = {n: e32.grads[n].detach().clone() for n in e32.grads}
  del e32
  torch.cuda.empty_cache()
  names, v16, e16 = step('bf16')
  got = gradient_stats(ref, {n: e16.grads[n].detach() for n in ref})
  l_ref = dict(zip([str(x) for x in g['loss_names']], g['losses_fp32']))
  l_ac = dict(zip([str(x) for x in g['loss_names']], g['losses_autocast']))
  l32 = dict(zip(names, v32))
  l16 = dict(zip(names, v16))
  loss_dev = {n: (abs(float(l16[n]) - l_ref[n]) / abs(l_ref[n]), abs(l_ac[n] - l_ref[n]) / abs(l_ref[n])) for n in names}
  _report('swin_bf16_bs4_vs_autocast_reference', {'hip_bf16_vs_hip_fp32': got, 'reference_autocast_vs_reference_fp32': want,
                                                  'losses_rel_dev_hip_and_autocast': {n: [float(a), float(b)] for n, (a, b) in loss_dev.items()}})
  for n in names:  # the fp32 step itself sits on the reference's fp32 losses
    assert abs(float(l32[n]) - l_ref[n]) <= 1e-3 * abs(l_ref[n]) + 1e-6, (n, l32[n], l_ref[n])
  slack = 1.02
  tail = {'norm_err_p90': 1.3, 'norm_err_p99': 1.4, 'norm_err_max': 1.5}
  assert got['arena_cosine'] >= 1.0 - (1.0 - want['arena_cosine']) * slack, (got, want)
  for k in STAT_KEYS[1:]:
    assert got[k] <= want[k] * tail.get(k, slack), (k, got, want)
  assert got['tensors'] >= 500
  # losses: twice the autocast reference's own distance from fp32, plus 1e-3 in LOSS UNITS for the small ones (loss_yaw_res = 0.033 is the mean
  # L1 residual of the handful of boxes of four samples: HIP bf16 0.0007 away in absolute terms = 2.1 %, the autocast reference 0.5 %)
  for n, (a, b) in loss_dev.items():
    assert a * abs(l_ref[n]) <= 2.0 * b * abs(l_ref[n]) + 1e-3 * max(1.0, abs(l_ref[n])), (n, a, b)


@pytest.mark.gpu
def test_bf16_trains_like_fp32_over_200_steps_and_drifts_no_further_than_the_autocast_reference():
  """Does the benchmarked precision TRAIN like fp32?  200 optimizer steps (AdamW-amsgrad, lr 1e-4, four different batches of 4 cycled, dropout
  off), a bf16 Trainer beside an fp32 Trainer from identical weights.  The bar is not an absolute number but the drift of a bf16 REFERENCE:
  tests/golden/tfpp_bf16_autocast_curve.npz holds the same 200 steps of the oracle port in fp32 and under torch.autocast(bfloat16)
  (oracle/make_golden_bf16_curve.py).  Compared on curves smoothed over 8 steps (two cycles of the four batches): these statistics move by
  +-30 % between two builds that only differ in the ORDER of fp32 sums (profiles/r05_bf16_curve.txt: smoothed max 2.9-3.8 %, mean 1.7-2.1 %
  over 200 steps), so the raw 50-step bars of rounds 3-4 (max 4 %, mean 1 %, set from one measurement) were inside their own noise."""
  from carla_garage_amd.trainer import Trainer
  g = U.load_golden('tfpp_bf16_autocast_curve.npz')
  steps = int(g['steps'])
  batches = []
  for i in range(4):
    b = {k: v.cuda() for k, v in P.make_labels(4).items()}
    for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(4)):
      b[k] = v.cuda()
    b['rgb'] = (b['rgb'] + 5.0 * i).clamp(0, 255)
    batches.append(b)
  curves = {}
  for dt in ('fp32', 'bf16'):
    m = _model(dt).train()
    _zero_dropout(m)
    tr = Trainer(m, lr=1e-4)
    curves[dt] = np.array([tr.total_loss(tr.train_step(batches[s % 4])) for s in range(steps)])
    del tr, m
    torch.cuda.empty_cache()
  smooth = lambda x: np.convolve(x, np.ones(8) / 8, mode='valid')
  dev = lambda x, ref: np.abs(smooth(x) - smooth(ref)) / np.abs(smooth(ref))
  a, b = curves['fp32'], curves['bf16']
  hip, ref = dev(b, a), dev(g['autocast'], g['fp32'])
  fp32_vs_ref = dev(a, g['fp32'])
  rep = {'steps': steps, 'fp32_first_last8': [float(a[0]), float(a[-8:].mean())], 'bf16_first_last8': [float(b[0]), float(b[-8:].mean())],
         'hip_bf16_vs_fp32_smoothed': {'max': float(hip.max()), 'mean': float(hip.mean())},
         'reference_autocast_vs_fp32_smoothed': {'max': float(ref.max()), 'mean': float(ref.mean())},
         'hip_fp32_vs_reference_fp32_smoothed': {'max': float(fp32_vs_ref.max()), 'mean': float(fp32_vs_ref.mean())},
         'reference_last8': [float(g['fp32'][-8:].mean()), float(g['autocast'][-8:].mean())]}
  _report('bf16_vs_fp32_200_steps', rep)
  assert np.isfinite(b).all() and a[-8:].mean() < 0.02 * a[0] and b[-8:].mean() < 0.02 * b[0], rep   # both train: 96.5 -> below 1.9
  assert abs(float(a[0]) - float(g['fp32'][0])) <= 1e-4 * float(a[0]), rep                            # the same first step as the oracle in fp32
  assert fp32_vs_ref.mean() <= 0.03, rep                                                                # ... and the same fp32 curve within its own chaos
  # the bf16 run drifts from its fp32 run no further than the reference's autocast run drifts from ITS fp32 run (smoothed: max 5.9 %, mean 3.1 %,
  # final loss 2.7 % lower) + half a percent / one percent for the order-of-sums noise.  Measured: max 2.9-3.8 %, mean 1.7-2.1 %, final loss 1.1-1.9 % lower
  assert hip.mean() <= ref.mean() + 0.005 and hip.max() <= ref.max() + 0.01, rep
  assert abs(b[-8:].mean() - a[-8:].mean()) <= 0.04 * a[-8:].mean(), rep


@pytest.mark.gpu
@pytest.mark.parametrize('path', ['rows', 'r5'])
def test_fused_batchnorm_backward_sums_do_not_change_the_bf16_step(monkeypatch, path):
  """The BatchNorm-backward sums emitted by the squeeze-excite backward kernel (the producer of d(y) for conv2 of every bottleneck,
  Engine.squeeze_excite) replace the separate reduction pass: the bf16 step with the fusion must reproduce the step without it (same rounded
  gradients go into the sums, only the summation order differs), and it must really be taken on the 42 conv2 BatchNorm layers.
  'rows': the round-6 path (conv2's output is a BnView, tfpp_se_bwd_apply_bn -> tfpp_bn_bwd_apply_rows2); 'r5': the launch sequence of rounds
  1-5 (engine.BN_ROWS off: tfpp_se_bwd_apply_bns -> tfpp_bn_bwd_apply_rows), kept for the configurations that still use it (SyncBatchNorm)."""
  from carla_garage_amd import engine as E
  from carla_garage_amd import ops
  monkeypatch.setattr(E, 'BN_ROWS', path == 'rows')
  out = {}
  calls = {}
  if path == 'r5':
    real_rows = ops.bn_bwd_rows
    for fused in (False, True):
      calls[fused] = [0]

      def count_rows(*a, _f=fused, **k):
        calls[_f][0] += 1
        return real_rows(*a, **k)

      monkeypatch.setattr(ops, 'bn_bwd_rows', count_rows)
      m = _model('bf16').train()
      with monkeypatch.context() as mp:
        if not fused:  # the producer only emits the sums when its gradient is the last one the tensor receives: say it never is
          mp.setattr(E.Tape, 'is_last_contribution', lambda self, t: False)
        names, vals, eng = _engine_train_step(m, 4)
      out[fused] = (vals, eng.flat_grad.detach().double().cpu().numpy().copy())
    monkeypatch.setattr(ops, 'bn_bwd_rows', real_rows)
  else:
    real_reduce, real_se = ops.bn_bwd_reduce_rows, ops.se_bwd_apply_bn
    for fused in (False, True):
      calls[fused] = [0]
      reduces = [0]

      def count_reduce(*a, _r=reduces, **k):
        _r[0] += 1
        return real_reduce(*a, **k)

      def se_apply(*a, _f=fused, **k):
        calls[_f][0] += 1
        dx, partial, nrows = real_se(*a, **k)
        return (dx, partial, nrows) if _f else (dx, None, 0)  # without the rows conv2's backward runs its own reduction pass

      monkeypatch.setattr(ops, 'bn_bwd_reduce_rows', count_reduce)
      monkeypatch.setattr(ops, 'se_bwd_apply_bn', se_apply)
      m = _model('bf16').train()
      names, vals, eng = _engine_train_step(m, 4)
      out[fused] = (vals, eng.flat_grad.detach().double().cpu().numpy().copy())
      calls[('reduce', fused)] = reduces[0]
    monkeypatch.setattr(ops, 'bn_bwd_reduce_rows', real_reduce)
    monkeypatch.setattr(ops, 'se_bwd_apply_bn', real_se)
    assert calls[('reduce', False)] - calls[('reduce', True)] == 42, calls  # the 42 reduction passes the fused sums replace
    calls[False][0] = 0  # (same bookkeeping as the r5 branch below: "fused launches used")
  assert calls[False][0] == 0 and calls[True][0] == 42, calls   # 21 bottlenecks x 2 encoders
  lerr = float(np.max(np.abs(out[True][0] - out[False][0]) / np.abs(out[False][0])))
  gerr = float(np.linalg.norm(out[True][1] - out[False][1]) / np.linalg.norm(out[False][1]))
  _report('fused_bn_bwd_' + path, {'fused_layers': calls[True][0], 'loss_rel': lerr, 'grad_rel_l2': gerr})
  assert lerr <= 1e-5 and gerr <= 2e-2, (lerr, gerr)  # losses: fp32 atomics of the loss sums; gradients: measured 4e-3


@pytest.mark.gpu
def test_eval_after_training_steps_uses_current_weights_and_running_statistics():
  """the fused optimizer and the BN running-statistic updates write through raw pointers, which tensor._version
  cannot see.  eval -> train steps -> eval must run on the updated weights and the re-folded running statistics: compared with a
  freshly constructed model that loads the trained state_dict."""
  from carla_garage_amd.trainer import Trainer
  m = _model('fp32')
  inp = [x.cuda() for x in P.make_inputs(1)]
  m.eval()
  with torch.inference_mode():
    o0 = m(*inp)
    before = [o0[1].clone(), o0[2].clone(), o0[6][0].clone()]
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()
  tr = Trainer(m, lr=1e-3)
  for _ in range(2):
    tr.train_step(batch)
  m.eval()
  with torch.inference_mode():
    out = m(*inp)
    after = [out[1], out[2], out[6][0]]
  fresh = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32'))
  fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, strict=True)
  fresh.cuda().eval()
  with torch.inference_mode():
    o2 = fresh(*inp)
    want = [o2[1], o2[2], o2[6][0]]
  torch.cuda.synchronize()
  for a, w, b in zip(after, want, before):
    assert U.rel_err(U.to_np(a), U.to_np(w)) <= 1e-5  # same kernels, same weights
    assert U.rel_err(U.to_np(a), U.to_np(b)) > 1e-4   # and it really changed with the two optimizer steps


@pytest.mark.gpu
def test_eval_forward_calls_are_captured_and_replayed_without_changing_results():
  """model.forward() as sensor_agent.py:456-461 calls it every tick (eval, inference_mode, the same shapes) with TFPP_EVAL_GRAPH_AFTER=2: the third call
  of a signature is captured into a hipGraph and later calls replay it (model.py _plain_forward).  Replayed results are bit-identical to the eager ones; the caller owns what it
  gets (a later call does not overwrite it); other inputs give other (correct) results; weights written in place, a loaded state_dict or training steps
  in between are picked up (the replay that read stale weight images is discarded); eval_graph_after = -1 keeps every call eager."""
  from carla_garage_amd.trainer import Trainer
  m = _model('fp32').eval()
  m.eval_graph_after = 2  # (what TFPP_EVAL_GRAPH_AFTER=2 in the agent's environment sets for every module)
  a = [x.cuda() for x in P.make_inputs(1)]
  b = [x.cuda() for x in P.make_inputs(1, seed=7)]
  pick = lambda o: [o[1], o[2], o[3], o[5], o[6][0], o[6][3]]
  with torch.inference_mode():
    first = pick(m(*a))
    assert m._eval_plans and all(pl['graph'] is None for pl in m._eval_plans.values())
    m(*a)
    third = pick(m(*a))  # captured here
    plan, = m._eval_plans.values()
    assert plan['graph'] is not None
    fourth = pick(m(*a))
    for x, y, z in zip(first, third, fourth):
      assert torch.equal(x, y) and torch.equal(x, z)
    other = pick(m(*b))  # the same signature, other values: a replay
  with torch.no_grad():  # (captured under inference_mode, replayed under no_grad: the graph's input buffers are ordinary tensors)
    again = pick(m(*a))
  with torch.inference_mode():
    for x, y, z, w in zip(first, fourth, other, again):
      assert torch.equal(x, y)  # what the caller got from the fourth call survived the two calls after it
      assert torch.equal(x, w) and not torch.equal(x, z)
    m.eval_graph_after = -1
    want_other = pick(m(*b))  # eager
    m.eval_graph_after = 2
    for z, w in zip(other, want_other):
      assert torch.equal(z, w)
    U.compare_packed(U.pack_outputs(m(*a)), U.load_golden('tfpp_eval_bs1.npz'))  # a replay against the reference's golden forward
  # weights written in place (outside inference_mode, where in-place writes do not advance tensor._version -- the eager path's repack check relies
  # on it just the same)
  with torch.no_grad():
    m.target_speed_network[2].bias.add_(0.5)
  with torch.inference_mode():
    moved = pick(m(*a))
    assert next(iter(m._eval_plans.values()))['graph'] is None  # the plan was dropped, this call ran eagerly on repacked weights
    torch.testing.assert_close(moved[0], first[0] + 0.5, rtol=0, atol=1e-5)
    assert torch.equal(moved[4], first[4])
    for _ in range(3):
      m(*a)
    assert next(iter(m._eval_plans.values()))['graph'] is not None
  # training steps in between (the fused optimizer writes the arena through raw pointers; BatchNorm statistics move)
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()
  tr = Trainer(m, lr=1e-3)
  for _ in range(2):
    tr.train_step(batch)
  m.eval()
  with torch.inference_mode():
    after = pick(m(*a))
  fresh = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32'))
  fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, strict=True)
  fresh.cuda().eval()
  with torch.inference_mode():
    want = pick(fresh(*a))
  for x, w, old in zip(after, want, moved):
    assert U.rel_err(U.to_np(x), U.to_np(w)) <= 1e-5 and U.rel_err(U.to_np(x), U.to_np(old)) > 1e-5


@pytest.mark.gpu
def test_the_streams_of_the_lanes_are_process_wide_and_pairwise_distinct():
  """torch.cuda.Stream() hands out 32 native streams per device round-robin: per-engine stream objects alias each other (and the capture stream)
  once a process has built a few models -- the cause of the round-5 crash inside hipGraphLaunch (carla_garage_amd/streams.py).  The registry
  hands every role its stream once, with pairwise distinct native handles, whatever the pool position is."""
  from carla_garage_amd import streams
  for _ in range(45):  # wrap the pool
    torch.cuda.Stream()
  got = [streams.get('cuda', 'capture'), streams.get('cuda', 'pack'), streams.get('cuda', 'comm'), streams.get('cuda', 'copy')]
  got += [streams.get('cuda', 'lane', k) for k in (1, 2)] + [streams.get('cuda', 'side', j) for j in range(4)]
  hs = [s.cuda_stream for s in got]
  assert len(set(hs)) == len(hs) and 0 not in hs, hs
  assert streams.get('cuda:0', 'lane', 1) is got[4] and streams.get(torch.device('cuda', 0), 'side', 3) is got[-1]
  m1, m2 = _model('bf16').eval(), _model('bf16').eval()
  inp = [x.cuda() for x in P.make_inputs(1)]
  with torch.inference_mode():
    a, b = m1(*inp)[2], m2(*inp)[2]
  assert torch.equal(a, b)
  assert m1._engine().lanes.branch is m2._engine().lanes.branch is streams.get('cuda', 'lane', 1)


@pytest.mark.gpu
def test_captured_eval_signatures_are_bounded(monkeypatch):
  """Every captured signature owns the activations of one forward: a module captures at most EVAL_GRAPH_MAX_PLANS of them, further signatures
  keep running eagerly.  Three batch sizes on one bf16 module, right behind the two tests above in the same process: the sequence that crashed
  inside hipGraphLaunch in round 5."""
  import carla_garage_amd.model as MM
  monkeypatch.setattr(MM, 'EVAL_GRAPH_MAX_PLANS', 2)
  m = _model('bf16').eval()
  m.eval_graph_after = 2
  ref = {}
  with torch.inference_mode():
    for bs in (1, 2, 3):
      inp = [x.cuda() for x in P.make_inputs(bs)]
      ref[bs] = m(*inp)[2].clone()
      for _ in range(4):
        got = m(*inp)[2]
      assert torch.equal(got, ref[bs])
    captured = sorted(k[1][0][0] for k, pl in m._eval_plans.items() if pl.get('graph') is not None)  # batch size of the rgb input of the signature
    assert captured == [1, 2], captured
    for bs in (3, 1, 2, 1):  # replays of the captured signatures and eager calls of the third one, interleaved
      inp = [x.cuda() for x in P.make_inputs(bs)]
      assert torch.equal(m(*inp)[2], ref[bs])


@pytest.mark.gpu
def test_trainer_state_dict_round_trip_and_reference_layout():
  """Trainer.state_dict() has the layout of the reference's optimizer_%04d.pth (torch.optim.AdamW(model.parameters(), amsgrad=True),
  team_code/train.py:529-534,967-976): torch's own AdamW loads it; save -> fresh trainer -> load -> the next step is identical."""
  from carla_garage_amd.trainer import Trainer
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()
  m = _model('fp32').train()
  _zero_dropout(m)
  tr = Trainer(m, lr=1e-4)
  tr.train_step(batch)
  tr.train_step(batch)
  torch.cuda.synchronize()
  osd = tr.state_dict()
  msd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
  n_train = sum(1 for p in m.parameters() if p.requires_grad)
  assert len(osd['state']) == n_train and osd['param_groups'][0]['params'] == list(range(len(list(m.parameters()))))
  probe = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(p, device='cpu')) for p in m.parameters()], lr=1.0, amsgrad=True)
  probe.load_state_dict({'state': {k: {a: b.cpu() for a, b in v.items()} for k, v in osd['state'].items()}, 'param_groups': osd['param_groups']})
  assert probe.param_groups[0]['lr'] == 1e-4
  snap = {a: getattr(tr, a).clone() for a in ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq', 'flat_param')}
  tr.train_step(batch)  # (the arenas move into the observed completion order at the start of this step: compare by parameter name below)
  torch.cuda.synchronize()
  by_name = lambda mod: torch.cat([p.detach().float().reshape(-1) for _, p in mod.named_parameters() if p.requires_grad])
  want, want_g = by_name(m).clone(), _grads_by_name(tr)
  m2 = LidarCenterNet(GlobalConfig(tfpp_dtype='fp32'))
  m2.load_state_dict(msd, strict=True)
  m2.cuda().train()
  _zero_dropout(m2)
  tr2 = Trainer(m2, lr=3e-4)
  tr2.load_state_dict(osd)
  assert tr2.step_count == 2 and tr2.lr == 1e-4
  for a, want_a in snap.items():  # the restored arenas are bit-identical to the ones the state was saved from
    assert torch.equal(getattr(tr2, a), want_a), a
  tr2.train_step(batch)
  torch.cuda.synchronize()
  gd = float(np.linalg.norm(_grads_by_name(tr2) - want_g) / np.linalg.norm(want_g))
  assert gd <= 1e-6, gd  # same weights, same batch: the third step's gradients agree to the fp32 run-to-run noise (5e-9 measured)
  # parameters: AdamW divides by sqrt(v): gradients that are pure rounding noise (structurally-zero key biases, pre-BN biases) turn
  # into updates of order lr with a run-dependent sign, so the parameters are compared to a fraction of one lr step, not to 1e-6
  d = (by_name(m2) - want).abs().max().item()
  assert d <= 0.5 * 1e-4, d


def _param_grads_vs_golden(m, g, tag, norm_tol_where=None):
  """``p.grad`` of every parameter (what the reference's optimizer reads after loss.backward()) against the reference's gradients: per-tensor
  norms and the sampled elements, at the bars of _check_train_step_vs_golden."""
  params = dict(m.named_parameters())
  worst = worst_el = 0.0
  over = {}
  for name, (norm, gmax), samples in zip(g['grad_names'], g['grad_norms'], g['grad_samples']):
    if gmax < 1e-5:
      continue
    mine = params[str(name)].grad.detach().flatten()
    e = abs(mine.double().norm().item() - norm) / norm
    worst = max(worst, e)
    bar = next((t for sub, t in (norm_tol_where or {}).items() if sub in str(name)), GRAD_NORM_TOL_SE_FC1 if '.se.fc' in str(name) else GRAD_NORM_TOL)
    if e > bar:  # (squeeze-excite layers 2e-2; norm_tol_where: tensors the calling test documents as ill-conditioned)
      over[str(name)] = e
    idx = U.sample_idx(mine.numel())
    got = mine[torch.from_numpy(idx).to(mine.device)].float().cpu().numpy()
    worst_el = max(worst_el, float(np.max(np.abs(got - samples[:len(idx)]) / (norm / np.sqrt(mine.numel()) + np.abs(samples[:len(idx)])))))
  _report(tag, {'worst_grad_norm': worst, 'worst_grad_elem': worst_el})
  assert not over and worst_el <= GRAD_ELEM_TOL, (over, worst_el)


@pytest.mark.gpu
def test_dropin_autograd_path_matches_engine_path():
  """forward -> compute_loss -> loss.backward() exactly as team_code/train.py:776-898 drives the model."""
  from carla_garage_amd.losses import normalized_loss_weights
  m = _model().train()
  _zero_dropout(m)
  lab = {k: v.cuda() for k, v in P.make_labels(2).items()}
  out = m(*[x.cuda() for x in P.make_inputs(2)])
  assert out[2].requires_grad and out[3].requires_grad
  losses = m.compute_loss(pred_wp=out[0], pred_target_speed=out[1], pred_checkpoint=out[2], pred_semantic=out[3],
                          pred_bev_semantic=out[4], pred_depth=out[5], pred_bounding_box=out[6], pred_wp_1=out[8], selected_path=out[9],
                          **lab)
  w = normalized_loss_weights(m.config)
  total = sum(w[k] * v for k, v in losses.items())
  total.backward()
  g = U.load_golden('tfpp_train_bs2.npz')
  np.testing.assert_allclose(float(total), float(g['total_loss']), rtol=1e-3)
  _param_grads_vs_golden(m, g, 'dropin')
  # predictions that are not the last forward's outputs (here: one clone) take compute_loss's general path (converted to the internal
  # layout) and give the same losses (tests/test_dropin_gpu.py covers gradients, accumulation, DDP and the hipGraph replays)
  with torch.no_grad():
    again = m.compute_loss(pred_wp=out[0], pred_target_speed=out[1].clone(), pred_checkpoint=out[2], pred_semantic=out[3],
                           pred_bev_semantic=out[4], pred_depth=out[5], pred_bounding_box=out[6], pred_wp_1=out[8], selected_path=out[9], **lab)
  for k in losses:
    np.testing.assert_allclose(float(again[k]), float(losses[k]), rtol=2e-5, err_msg=k)


def _grads_by_name(tr):
  """the gradient arena in model.named_parameters() order (the arena itself is laid out by completion order, which differs between lane set-ups)"""
  return np.concatenate([tr.eng.grads[n].detach().double().cpu().numpy().ravel() for n, p in tr.model.named_parameters() if p.requires_grad])


@pytest.mark.gpu
def test_streams_and_hipgraph_do_not_change_the_training_step():
  """Identical fp32 trainers run four steps on one stream / on the concurrent lanes / lanes + hipGraph replay of the fourth: losses and
  gradients must agree to the run-to-run noise of the single-stream path itself (order of the fp32 atomics of the fused BatchNorm
  statistics: ~1e-6 on the losses, ~2e-3 relative L2 on the gradients of step 1; a race between streams shows up orders of magnitude
  above that).  Steps 2-3 are where the arenas move into the observed completion order (Trainer.apply_observed_layout): parameters and
  optimizer state have to survive the move.  bf16 is not used here: with train-mode BN at batch 2 its run-to-run gradient noise is 0.2
  relative L2 on one stream already (tools/stress_step.py, phase B)."""
  from carla_garage_amd.graph import GraphedTrainStep
  from carla_garage_amd.trainer import Trainer
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()
  saved = {k: os.environ.get(k) for k in ('TFPP_BRANCH_STREAMS', 'TFPP_SIDE_STREAM')}

  def run(single, graph):
    for k in ('TFPP_BRANCH_STREAMS', 'TFPP_SIDE_STREAM'):
      if single:
        os.environ[k] = '0'
      else:
        os.environ.pop(k, None)
    m = _model('fp32').train()
    _zero_dropout(m)
    tr = Trainer(m, lr=1e-5)
    assert tr.eng.lanes.enabled == (not single) and tr.eng.side.enabled == (not single)
    v1 = tr.train_step(batch).detach().float().cpu().numpy().copy()
    g1 = _grads_by_name(tr)
    tr.train_step(batch)
    tr.train_step(batch)
    assert tr.layout_final and tr.eager_steps_in_layout >= 1  # moved at the start of step 3 (or found in place)
    if not single:
      assert len(tr.eng.buckets.ranges()) >= 3 and len(tr.program[1]) >= 2, (tr.eng.buckets.ranges(), tr.program, tr.eng.buckets.poisoned)
    if graph:
      gs = GraphedTrainStep(tr, batch, warmup=0)
      assert tr.step_count == 3  # (ready: no further eager step in front of the capture)
      v4 = gs(batch).detach().float().cpu().numpy().copy()
    else:
      v4 = tr.train_step(batch).detach().float().cpu().numpy().copy()
    torch.cuda.synchronize()
    assert tr.eng.buckets.timed_out() == 0
    return v1, g1, v4, _grads_by_name(tr)

  try:
    ref = run(True, False)
    assert all(np.isfinite(a).all() for a in ref)
    errs = {}
    for name, args in (('lanes', (False, False)), ('lanes+graph', (False, True))):
      r = run(*args)
      errs[name] = {'loss1': float(np.max(np.abs(r[0] - ref[0]) / np.abs(ref[0]))), 'grad1': float(np.linalg.norm(r[1] - ref[1]) / np.linalg.norm(ref[1])),
                    'loss4': float(np.max(np.abs(r[2] - ref[2]) / np.abs(ref[2]))), 'grad4': float(np.linalg.norm(r[3] - ref[3]) / np.linalg.norm(ref[3]))}
  finally:
    for k, v in saved.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v
  _report('streams', errs)
  for name, e in errs.items():
    assert e['loss1'] < 1e-4 and e['grad1'] < 2e-2 and e['loss4'] < 1e-2 and e['grad4'] < 1e-1, (name, e)


@pytest.mark.gpu
def test_completion_signals_carry_their_pass_and_a_timeout_raises_at_the_next_step(monkeypatch):
  """ADVICE r4 (two mediums) on carla_garage_amd/buckets.py.  (1) A signal carries the serial number of the pass that raised it: a pass the
  host never book-kept (a bare graph.replay(), an eager pass that died) re-raises an OLD serial and cannot release the next exchange early --
  the wait stays pending until the book-kept pass raises its own.  (2) A wait that gives up is reported at the NEXT exchange (the time-out
  word travels to pinned host memory behind the step's collectives) -- in Trainer.finish_step and DropinStep._run_backward, not only at a
  checkpoint."""
  import time
  from carla_garage_amd import buckets as B
  from carla_garage_amd import dist as tdist
  monkeypatch.setattr(tdist, 'exchange_enabled', lambda group=None: True)
  monkeypatch.setattr(tdist, 'world_size', lambda group=None: 1)

  class Done:

    def wait(self):
      pass

  monkeypatch.setattr(tdist, 'all_reduce_async', lambda t, group=None, avg=False: Done())
  flat = torch.zeros(256, device='cuda')
  # ---- (1) stale serials never release a later wait
  bk = B.GradBuckets()
  bk.configure([0, 128, 256], 'cuda', observed=True)
  bk.begin_issue()
  bk.begin_pass()
  prog = bk.finish()                      # pass 1 raises the end marker with serial 1
  bk.exchange(flat, prog)
  torch.cuda.synchronize()
  bk.begin_pass()
  bk.finish()                             # a pass nobody book-kept: re-raises serial 1
  bk.begin_issue()                        # pass 2 is "issued" but its signal nodes have not run yet
  bk.exchange(flat, prog)                 # waits for serial 2 on the exchange stream
  time.sleep(0.2)
  assert not bk.comm.query(), 'the wait for pass 2 was released by the stale signal of an un-book-kept pass'
  bk.begin_pass()
  bk.finish()                             # now pass 2 raises its end marker
  torch.cuda.synchronize()
  assert bk.comm.query() and bk.timed_out() == 0
  bk.raise_if_timed_out(block=True)
  # ---- (2) a wait that gives up raises at the next exchange
  monkeypatch.setattr(B, 'WAIT_TIMEOUT_MS', 50)
  bk2 = B.GradBuckets()
  bk2.configure([0, 128, 256], 'cuda', observed=True)
  bk2.begin_issue()
  bk2.exchange(flat, ((), ()))            # nothing ever raises serial 1: the wait gives up after 50 ms, the "all-reduce" runs on whatever is there
  torch.cuda.synchronize()
  assert bk2.timed_out() == 1
  bk2.begin_issue()
  bk2.begin_pass()
  prog2 = bk2.finish()
  with pytest.raises(RuntimeError, match='completion-signal wait'):
    bk2.exchange(flat, prog2)
  bk2.exchange(flat, prog2)               # reported once; training code that catches it can go on
  torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'hipgraph'])
def test_gradient_buckets_are_complete_when_their_signal_fires(graph, monkeypatch):
  """The overlapped gradient exchange (buckets.py) starts the all-reduce of a bucket behind that bucket's completion signal, while backward
  is still running.  Here the "collective" is a snapshot of the bucket taken on the exchange stream right behind the signal wait: every
  snapshot must equal the bucket as it stands when the step has ended -- bit for bit, eager and replayed from the hipGraph -- and the
  early buckets must really be early (their snapshot is done well before the pass ends)."""
  from carla_garage_amd import dist as tdist
  from carla_garage_amd.graph import GraphedTrainStep
  from carla_garage_amd.trainer import Trainer
  batch = {k: v.cuda() for k, v in P.make_labels(2).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(2)):
    batch[k] = v.cuda()
  m = _model('bf16').train()
  tr = Trainer(m, lr=0.0)
  for _ in range(3):
    tr.train_step(batch)
  assert tr.layout_final and len(tr.eng.buckets.ranges()) >= 3
  snaps = []

  class Done:
    def wait(self):
      pass

  def snapshot(t, group=None, avg=False):  # runs on the exchange stream, ordered behind the bucket's signal wait only
    ev = torch.cuda.Event(enable_timing=True)
    snaps.append((t, t.clone(), ev))
    ev.record()
    return Done()

  monkeypatch.setattr(tdist, 'exchange_enabled', lambda group=None: True)
  monkeypatch.setattr(tdist, 'all_reduce_async', snapshot)
  step = GraphedTrainStep(tr, batch, warmup=0) if graph else (lambda b: tr.train_step(b))
  for it in range(3):
    del snaps[:]
    end = torch.cuda.Event(enable_timing=True)
    step(batch)
    end.record()
    torch.cuda.synchronize()
    prog = step.program if graph else tr.program
    assert tr.eng.buckets.poisoned is None and len(prog[1]) >= 2, (prog, tr.eng.buckets.poisoned)
    assert len(snaps) == len(tr.eng.buckets.ranges())
    for b, (live, snap, ev) in enumerate(snaps):
      assert float(snap.abs().sum()) > 0 and torch.equal(live, snap), f'bucket {b} changed after its completion signal (iteration {it})'
    lead = snaps[0][2].elapsed_time(end)
    assert lead > 0.3, f'the first bucket was exchanged only {lead:.3f} ms before the step ended: not overlapped'
  assert tr.eng.buckets.timed_out() == 0
  _report('bucket_signals_' + ('graph' if graph else 'eager'), {'buckets': len(snaps), 'early_signals': len(prog[1]), 'first_bucket_lead_ms': lead,
                                                               'bucket_mb': [round(4e-6 * (hi - lo), 1) for lo, hi in tr.eng.buckets.ranges()]})
