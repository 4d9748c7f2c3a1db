"""Import the REAL reference (``/root/reference/team_code``) unmodified on CPU.

Test infrastructure (see oracle/__init__.py).  Works only in the build
container, where ``/root/reference`` exists; used to (a) validate
``oracle.tfpp_port`` and (b) generate tests/golden fixtures
(oracle/make_golden.py).  Nothing that runs on the GPU box calls this.

Recipe: inert ``sys.modules`` stubs for the
simulator / image-IO packages the model never uses at run time, ``np.string_``
shim (team_code/data.py:212-226 runs inside ``LidarCenterNet.__init__`` via
team_code/model.py:33), and ``oracle.timm_regnet`` registered as ``timm``.
"""
import os
import sys
import types
import importlib.machinery
import numpy as np
import torch  # noqa: F401  (fully import torch before any stub module exists)

REF_ROOT = '/root/reference'
REF_TEAM_CODE = os.path.join(REF_ROOT, 'team_code')


def available():
  return os.path.isdir(REF_TEAM_CODE)


class _Inert:
  """Object that absorbs any attribute access / call (stub for unused APIs)."""

  def __init__(self, *a, **k):
    pass

  def __call__(self, *a, **k):
    return _Inert()

  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _Inert()


def _stub(name, **attrs):
  if name in sys.modules and not getattr(sys.modules[name], '_tfpp_stub', False):
    return sys.modules[name]
  m = types.ModuleType(name)
  m.__spec__ = importlib.machinery.ModuleSpec(name, None)
  m._tfpp_stub = True

  def _module_getattr(attr):  # PEP 562 module-level __getattr__
    if attr.startswith('__'):
      raise AttributeError(attr)
    return _Inert()

  m.__getattr__ = _module_getattr
  for k, v in attrs.items():
    setattr(m, k, v)
  sys.modules[name] = m
  return m


def install_stubs():
  import transformers  # noqa: F401  (import BEFORE a fake torchvision appears)
  if not hasattr(np, 'string_'):
    np.string_ = np.bytes_
  for name in ('carla', 'cv2', 'ujson', 'laspy', 'shapely', 'shapely.geometry', 'imgaug', 'imgaug.augmenters',
               'torchvision', 'torchvision.models', 'torchvision.models.video'):
    try:
      if name not in sys.modules:
        __import__(name)
    except Exception:  # pylint: disable=broad-except
      _stub(name)
  if getattr(sys.modules.get('imgaug'), '_tfpp_stub', False):
    sys.modules['imgaug'].augmenters = sys.modules['imgaug.augmenters']
  if getattr(sys.modules.get('shapely'), '_tfpp_stub', False):
    sys.modules['shapely'].geometry = sys.modules['shapely.geometry']
  from oracle import timm_regnet
  timm_regnet.install_as_timm()
  if REF_TEAM_CODE not in sys.path:
    sys.path.insert(0, REF_TEAM_CODE)


_cache = {}


def reference_modules():
  """Returns (config_module, model_module) of the unmodified reference."""
  if not available():
    raise RuntimeError('reference tree not present (expected only in the build container)')
  if 'mods' not in _cache:
    install_stubs()
    import config as ref_config  # pylint: disable=import-error
    import model as ref_model  # pylint: disable=import-error
    _cache['mods'] = (ref_config, ref_model)
  return _cache['mods']


def build_reference_model(**overrides):
  """``LidarCenterNet(GlobalConfig())`` of the reference with config attribute overrides."""
  ref_config, ref_model = reference_modules()
  cfg = ref_config.GlobalConfig()
  for k, v in overrides.items():
    setattr(cfg, k, v)
  return ref_model.LidarCenterNet(cfg), cfg
