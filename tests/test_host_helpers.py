"""Host-side helpers of the boundary module against the LIVE reference (build container only; -m "not gpu"): the PID controllers
(team_code/model.py:461-554, transfuser_utils.py:316-338), the optimizer grouping (model.py:556-632) and the box conversion behind
convert_features_to_bb_metric (model.py:447-459, transfuser_utils.py:388-406).  These run on the CPU in the reference too, so the
replacement must be bit-identical in behaviour."""
import numpy as np
import pytest
import torch

from oracle import ref_harness
from carla_garage_amd.config import GlobalConfig
from carla_garage_amd.model import LidarCenterNet

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason='needs /root/reference (build container only)')


@pytest.fixture(scope='module')
def pair():
  ref, _ = ref_harness.build_reference_model()
  return ref, LidarCenterNet(GlobalConfig())


def test_control_pid_direct_matches_reference_over_a_random_drive(pair):
  ref, mine = pair
  rng = np.random.RandomState(0)
  for step in range(300):
    ts = float(rng.choice([0.0, 0.005, 2.0, 5.0, 8.0])) if step % 7 else 0.0
    angle = float(rng.uniform(-1.2, 1.2))
    speed = torch.tensor([float(rng.choice([0.0, 0.005, 1.0, 4.0, 9.0]))])  # gt_velocity, shape (1,) (sensor_agent.py:434,555)
    a, b = ref.control_pid_direct(ts, angle, speed), mine.control_pid_direct(ts, angle, speed)
    assert a[0] == b[0] and float(a[1]) == float(b[1]) and bool(a[2]) == bool(b[2]), (step, a, b)


def test_control_pid_matches_reference_over_random_waypoints(pair):
  ref, mine = pair
  rng = np.random.RandomState(1)
  for step in range(300):
    wp = torch.from_numpy(np.cumsum(rng.uniform(-0.2, 1.5, size=(1, 8, 2)), axis=1).astype(np.float32))
    vel = torch.tensor([float(rng.choice([0.0, 0.005, 1.0, 4.0, 9.0]))])
    a, b = ref.control_pid(wp, vel), mine.control_pid(wp, vel)
    assert float(a[0]) == float(b[0]) and float(a[1]) == float(b[1]) and bool(a[2]) == bool(b[2]), (step, a, b)


def test_optimizer_groups_equal_the_reference_partition(pair):
  ref, mine = pair
  names_r = {id(p): n for n, p in ref.named_parameters()}
  names_m = {id(p): n for n, p in mine.named_parameters()}
  gr, gm = ref.create_optimizer_groups(0.01), mine.create_optimizer_groups(0.01)
  for a, b in zip(gr, gm):
    assert a['weight_decay'] == b['weight_decay']
    assert sorted(names_r[id(p)] for p in a['params']) == sorted(names_m[id(p)] for p in b['params'])


def test_convert_features_to_bb_metric_conversion_matches_reference(pair, monkeypatch):
  ref, mine = pair
  import transfuser_utils as t_u  # the reference's, on sys.path through ref_harness
  rng = np.random.RandomState(2)
  boxes = rng.uniform(0, 256, size=(1, 100, 9)).astype(np.float32)
  boxes[0, :, 4] = rng.uniform(-3.2, 3.2, 100)
  boxes[0, :, 8] = rng.uniform(0, 1, 100)  # confidence
  monkeypatch.setattr(mine.head, 'get_bboxes', lambda *a: torch.from_numpy(boxes.copy()))
  got = mine.convert_features_to_bb_metric([None] * 7)
  keep = boxes[0][boxes[0, :, -1] > ref.config.bb_confidence_threshold]
  want = [t_u.bb_image_to_vehicle_system(b.copy(), ref.config.pixels_per_meter, ref.config.min_x, ref.config.min_y) for b in keep]
  assert len(got) == len(want) > 10
  for a, b in zip(got, want):
    np.testing.assert_array_equal(a, b)


def test_no_decay_bitmask_marks_whole_parameters_and_their_padding():
  """ops.no_decay_bitmask (host side of tfpp_adamw_amsgrad_groups): one bit per group of 4 arena elements, little-endian inside 32-bit words."""
  import numpy as np
  from carla_garage_amd import ops
  slices = [(0, 6, 'a'), (8, 4, 'b'), (12, 130, 'c'), (144, 3, 'd'), (148, 1000, 'e')]
  total = 148 + 1000
  words = ops.no_decay_bitmask(slices, {'b', 'd', 'e'}, total)
  assert words.dtype == np.int32 and words.size == ((total + 3) // 4 + 31) // 32
  bits = np.unpackbits(words.view(np.uint8), bitorder='little')
  want = np.zeros_like(bits)
  for off, n, name in slices:
    if name in ('b', 'd', 'e'):
      want[off // 4:(off + n + 3) // 4] = 1
  assert np.array_equal(bits, want)
  assert not ops.no_decay_bitmask(slices, set(), total).any()


def test_the_product_never_imports_the_oracle():
  """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() (tools/smoke_check.py, the checker of one small invocation) and
  bench.py's cpu_baseline / bit-exactness legs may import it; no file of the package is exempt.  Static check over the package sources + a dynamic one: importing every product
  module leaves no `oracle` module behind."""
  import glob
  import os
  import re
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  offenders = []
  for f in glob.glob(os.path.join(root, 'carla_garage_amd', '*.py')):
    if re.search(r'^\s*(from|import)\s+oracle\b', open(f, encoding='utf-8').read(), re.M):
      offenders.append(f)
  assert not offenders, offenders
  mods = sorted(os.path.basename(f)[:-3] for f in glob.glob(os.path.join(root, 'carla_garage_amd', '*.py')) if os.path.basename(f) != '__init__.py')
  code = 'import sys, importlib\n' + ''.join(f'importlib.import_module("carla_garage_amd.{m}")\n' for m in mods) + \
      'bad = [k for k in sys.modules if k == "oracle" or k.startswith("oracle.")]\nassert not bad, bad\nprint("ok")'
  r = subprocess.run([sys.executable, '-c', code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, check=False)
  assert r.returncode == 0 and r.stdout.decode().strip().endswith('ok'), r.stdout.decode()[-2000:]
