#!/usr/bin/env python
"""Launcher shim: runs the reference's OWN ``team_code/train.py`` unmodified (autonomousvision/carla_garage @ 2024_08_07).

    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 \\
    python tools/reference_train_shim.py --reference /path/to/carla_garage [--mi355x] [--cpu] [--synthetic N] -- <train.py arguments>

SURVEY.md section 7 ("train.py cannot run unmodified on this stack") lists why a shim is needed; it does exactly that and nothing
else -- no line of the reference is edited or copied:
  * ``MultiStepLR(..., verbose=True)`` / ``CosineAnnealingWarmRestarts(..., verbose=False)`` (train.py:589-598): the ``verbose``
    keyword no longer exists in torch 2.10 -> accepted and dropped;
  * ``diskcache``, ``torchmetrics``, ``tensorboard`` and the simulator / image-IO packages are not installed -> inert stubs
    (oracle/ref_harness.py installs the same ones for the model code);
  * ``--synthetic N``: ``CARLA_Data`` is subclassed so that it serves N seeded synthetic frames of the shapes of SURVEY.md 8(d) instead of
    reading a dataset (there is none here);
  * ``--cpu``: BASELINE config 1 is specified on the CPU (AIM backbone, bs = 2, 10 frames, no GPU).  train.py hard-codes
    ``torch.device('cuda:N')``, ``backend='nccl'`` and ``model.cuda()`` (train.py:359-361,483): the name ``torch`` that train.py imports
    resolves to a forwarding proxy whose ``device('cuda:N')`` is the CPU device, ``init_process_group`` uses gloo, ``Module.cuda`` is a
    no-op;
  * ``--mi355x``: the one-line swap of INTEGRATION.md -- ``model.LidarCenterNet`` is replaced by ``carla_garage_amd.model.LidarCenterNet``
    before train.py executes ``from model import LidarCenterNet`` (train.py:32).  Everything else (argparse -> GlobalConfig, DDP wrap,
    ZeroRedundancyOptimizer / AdamW, schedulers, the epoch loop, checkpoint files) is the reference's code;
  * ``--fused-optimizer``: the optional second substitution of INTEGRATION.md -- ``optim.AdamW`` / ``ZeroRedundancyOptimizer`` (train.py:527-531)
    resolve to ``carla_garage_amd.optim.FlatAdamW`` (bench.py 'dropin': 31.1 instead of 34.3 ms/step).
"""
import argparse
import os
import runpy
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def _patch_schedulers():
  import torch
  for name in ('MultiStepLR', 'CosineAnnealingWarmRestarts'):
    cls = getattr(torch.optim.lr_scheduler, name)
    init = cls.__init__
    if getattr(init, '_tfpp_shim', False):
      continue

    def wrapped(self, *a, _init=init, **k):
      k.pop('verbose', None)
      return _init(self, *a, **k)

    wrapped._tfpp_shim = True
    cls.__init__ = wrapped


def _stub_modules():
  from oracle import ref_harness
  ref_harness.install_stubs()  # carla, cv2, ujson, laspy, shapely, imgaug, torchvision, timm (restated RegNet), np.string_
  if 'diskcache' not in sys.modules:
    m = types.ModuleType('diskcache')
    m.Cache = lambda *a, **k: {}
    sys.modules['diskcache'] = m
  if 'torchmetrics' not in sys.modules:
    import torch
    tm = types.ModuleType('torchmetrics')
    fn = types.ModuleType('torchmetrics.functional')
    fn.jaccard_index = lambda *a, **k: torch.tensor(0.0)
    tm.functional = fn
    sys.modules['torchmetrics'], sys.modules['torchmetrics.functional'] = tm, fn
  try:
    import torch.utils.tensorboard  # noqa: F401
  except Exception:  # pylint: disable=broad-except
    import torch.utils as tu
    tb = types.ModuleType('torch.utils.tensorboard')

    class SummaryWriter:

      def __init__(self, *a, **k):
        pass

      def add_scalar(self, *a, **k):
        pass

      def close(self):
        pass

    tb.SummaryWriter = SummaryWriter
    sys.modules['torch.utils.tensorboard'] = tb
    tu.tensorboard = tb


def _synthetic_dataset(n):
  """Subclass of the reference's CARLA_Data (its __init__ runs with an empty root list, as LidarCenterNet.__init__ itself does at
  model.py:33) that returns n seeded synthetic frames with the keys and dtypes Engine.load_data_compute_loss reads (train.py:688-766)."""
  import numpy as np
  import data as ref_data  # the reference's module, on sys.path
  base = ref_data.CARLA_Data
  if getattr(base, '_tfpp_synthetic', False):
    return

  class SyntheticCarlaData(base):
    _tfpp_synthetic = True

    def __init__(self, root, config, *a, **k):
      super().__init__(root=[], config=config)
      self._n = n if isinstance(root, (list, tuple)) and (len(root) > 0 or k.get('rank') is not None or len(a) > 0) else 0
      self._cfg = config

    def __len__(self):
      return self._n

    def __getitem__(self, index):
      c = self._cfg
      rng = np.random.RandomState(1234 + index)
      hb, wb = c.lidar_resolution_height // c.bev_down_sample_factor, c.lidar_resolution_width // c.bev_down_sample_factor
      d = {
          'rgb': rng.randint(0, 256, (3, c.camera_height, c.camera_width)).astype(np.uint8),
          'lidar': ((rng.rand(1, c.lidar_resolution_height, c.lidar_resolution_width) < 0.1) * rng.randint(1, 6, (1, c.lidar_resolution_height, c.lidar_resolution_width)) / 5.0).astype(np.float32),
          'target_point': (rng.randn(2) * np.array([20.0, 5.0])).astype(np.float32),
          'command': np.eye(6, dtype=np.float32)[rng.randint(0, 6)],
          'speed': np.float32(rng.rand() * 8.0),
          'target_speed': np.int64(rng.randint(0, len(c.target_speeds))),
          'route': (rng.randn(20, 2) * 5).astype(np.float32),
          'ego_waypoints': (rng.randn(c.pred_len, 2) * 5).astype(np.float32),
          'semantic': rng.randint(0, c.num_semantic_classes, (c.camera_height, c.camera_width)).astype(np.int64),
          'bev_semantic': rng.randint(0, c.num_bev_semantic_classes, (c.lidar_resolution_height, c.lidar_resolution_width)).astype(np.int64),
          'depth': rng.rand(c.camera_height, c.camera_width).astype(np.float32),
          'bounding_boxes': np.zeros((c.max_num_bbs, 8), np.float32),
          'center_heatmap': np.zeros((c.num_bb_classes, hb, wb), np.float32),
          'wh': np.zeros((2, hb, wb), np.float32), 'yaw_class': np.zeros((hb, wb), np.int64), 'yaw_res': np.zeros((1, hb, wb), np.float32),
          'offset': np.zeros((2, hb, wb), np.float32), 'velocity': np.zeros((1, hb, wb), np.float32), 'brake_target': np.zeros((hb, wb), np.int64),
          'pixel_weight': np.zeros((2, hb, wb), np.float32), 'avg_factor': np.float32(1.0),
      }
      d['center_heatmap'][0, hb // 2, wb // 2] = 1.0
      d['pixel_weight'][:, hb // 2, wb // 2] = 1.0
      return d

  ref_data.CARLA_Data = SyntheticCarlaData


def _cpu_torch_proxy():
  """sys.modules['torch'] := a module that forwards every attribute to the real torch except the few CUDA-only calls train.py makes by
  name.  Only code that executes ``import torch`` AFTER this point sees it (train.py and the team_code modules it imports); torch itself
  keeps its own references."""
  import torch as real
  import torch.distributed as rdist

  class _DeviceMeta(type):

    def __instancecheck__(cls, obj):
      return isinstance(obj, real.device)

  class device(metaclass=_DeviceMeta):  # torch.device('cuda:0') -> the CPU device

    def __new__(cls, spec='cpu', *a):
      return real.device('cpu') if 'cuda' in str(spec) else real.device(spec, *a)

  class _NullCtx:

    def __init__(self, *a, **k):
      pass

    def __enter__(self):
      return self

    def __exit__(self, *a):
      return False

  cuda = types.ModuleType('torch.cuda')
  cuda.__dict__.update({k: getattr(real.cuda, k) for k in dir(real.cuda) if not k.startswith('__')})
  cuda.device_count = lambda: 1
  cuda.device = _NullCtx
  cuda.empty_cache = lambda: None
  cuda.is_available = lambda: False

  dist = types.ModuleType('torch.distributed')
  dist.__dict__.update({k: getattr(rdist, k) for k in dir(rdist) if not k.startswith('__')})

  def init_process_group(backend=None, **k):  # 'nccl' needs GPUs; the CPU configuration runs the same collectives on gloo
    return rdist.init_process_group(backend='gloo', **k)

  dist.init_process_group = init_process_group

  class Proxy(types.ModuleType):

    def __getattr__(self, name):
      return getattr(real, name)

  proxy = Proxy('torch')
  proxy.__dict__.update(device=device, cuda=cuda, distributed=dist, __path__=real.__path__, __file__=real.__file__, __spec__=real.__spec__,
                        __version__=real.__version__)
  real.nn.Module.cuda = lambda self, device=None: self  # model.cuda(device=...) at train.py:483
  sys.modules['torch'] = proxy
  return real


def main():
  ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  ap.add_argument('--reference', default='/root/reference', help='checkout of autonomousvision/carla_garage')
  ap.add_argument('--cpu', action='store_true', help='BASELINE config 1: run on the CPU over gloo')
  ap.add_argument('--mi355x', action='store_true', help='swap in carla_garage_amd.model.LidarCenterNet (INTEGRATION.md)')
  ap.add_argument('--fused-optimizer', action='store_true', help='substitute carla_garage_amd.optim.FlatAdamW for optim.AdamW / ZeroRedundancyOptimizer')
  ap.add_argument('--synthetic', type=int, default=0, help='serve N synthetic frames instead of a dataset')
  args, rest = ap.parse_known_args()
  if rest and rest[0] == '--':
    rest = rest[1:]
  team_code = os.path.join(args.reference, 'team_code')
  if not os.path.isdir(team_code):
    sys.exit(f'{team_code} not found (pass --reference)')
  os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
  from oracle import ref_harness
  ref_harness.REF_ROOT, ref_harness.REF_TEAM_CODE = args.reference, team_code
  _stub_modules()
  _patch_schedulers()
  if args.synthetic > 0:
    _synthetic_dataset(args.synthetic)
    if '--root_dir' not in rest:
      # GlobalConfig.initialize (config.py:546-598) lists <root_dir>/<scenario>/<Town.._RepetitionN> directories: one empty route
      # directory satisfies the scan; the synthetic dataset never opens it
      import tempfile
      root = tempfile.mkdtemp(prefix='tfpp_synth_')
      os.makedirs(os.path.join(root, 'synthetic', 'Town01_Repetition0'))
      rest += ['--root_dir', root]
  if args.mi355x:
    import model as ref_model  # the reference's module (sys.path), imported for the swap only
    from carla_garage_amd.model import LidarCenterNet
    ref_model.LidarCenterNet = LidarCenterNet
    print('[shim] model.LidarCenterNet -> carla_garage_amd.model.LidarCenterNet', flush=True)
  if args.fused_optimizer:
    # the second line of INTEGRATION.md: train.py:527-531 builds optim.AdamW(params, lr, amsgrad=True), optionally inside
    # ZeroRedundancyOptimizer; both names resolve to carla_garage_amd.optim.FlatAdamW (one fused launch over the flat arenas; ZeRO-1
    # sharding of 1.9 GB of state is pointless at 288 GB per GPU)
    import torch.optim
    import torch.distributed.optim as tdo
    from carla_garage_amd.optim import FlatAdamW
    torch.optim.AdamW = FlatAdamW
    tdo.ZeroRedundancyOptimizer = lambda params, optimizer_class=None, **kw: FlatAdamW(params, **kw)
    print('[shim] optim.AdamW / ZeroRedundancyOptimizer -> carla_garage_amd.optim.FlatAdamW', flush=True)
  if args.cpu:
    _cpu_torch_proxy()
  sys.argv = [os.path.join(team_code, 'train.py')] + rest
  runpy.run_path(sys.argv[0], run_name='__main__')


if __name__ == '__main__':
  main()
