"""The reference's OWN team_code/train.py driven through tools/reference_train_shim.py (build container only, -m "not gpu").

BASELINE config 1: AIM image-only backbone, bs = 2, 10 synthetic 256x1024 frames, reference train.py on the CPU (plumbing, no GPU) --
and the one-line import swap of INTEGRATION.md: the same train.py builds carla_garage_amd.model.LidarCenterNet from its own
argparse -> GlobalConfig, wraps it in DistributedDataParallel, creates the ZeRO AdamW, the schedulers and the data loader, and
reaches the first forward, where the MI355X module refuses CPU tensors (there is no GPU in this container; on a GPU host the same
command trains)."""
import os
import socket
import subprocess
import sys

import pytest

from oracle import ref_harness

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not ref_harness.available(), reason='needs /root/reference (build container only)')


def _run(tmp_path, extra, train_args):
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  cmd = [sys.executable, os.path.join(ROOT, 'tools', 'reference_train_shim.py'), '--cpu', '--synthetic', '10'] + extra + ['--'] + train_args + [
      '--batch_size', '2', '--epochs', '1', '--cpu_cores', '1', '--logdir', str(tmp_path), '--use_disk_cache', '0', '--setting', 'all']
  p = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, check=False)
  return p.returncode, p.stdout.decode()


def test_baseline_config_1_reference_train_py_runs_aim_on_cpu(tmp_path):
  rc, out = _run(tmp_path, [], ['--id', 'aim', '--backbone', 'aim', '--use_semantic', '0', '--use_depth', '0', '--detect_boxes', '0',
                                '--use_bev_semantic', '0'])
  assert rc == 0, out[-3000:]
  assert 'Total trainable parameters:  27949776' in out   # RegNetY-3.2GF image branch + planning head
  assert '5/5' in out                                     # 10 frames / bs 2 = 5 optimizer steps
  files = sorted(os.listdir(os.path.join(str(tmp_path), 'aim')))
  assert files == ['args.txt', 'config.pickle', 'model_0000.pth', 'optimizer_0000.pth', 'scaler_0000.pth', 'scheduler_0000.pth']
  import torch
  sd = torch.load(os.path.join(str(tmp_path), 'aim', 'model_0000.pth'), map_location='cpu')
  # the checkpoint the reference wrote loads into the MI355X module of the same configuration (strict: identical key set / shapes)
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  m = LidarCenterNet(GlobalConfig(backbone='aim', use_semantic=0, use_depth=0, detect_boxes=0, use_bev_semantic=0))
  m.load_state_dict(sd, strict=True)


def test_import_swap_lets_reference_train_py_build_and_wrap_the_mi355x_module(tmp_path):
  rc, out = _run(tmp_path, ['--mi355x'], ['--id', 'swap', '--backbone', 'transFuser'])
  assert '[shim] model.LidarCenterNet -> carla_garage_amd.model.LidarCenterNet' in out
  assert 'Total trainable parameters:  120219954' in out  # train.py:477-481 counted OUR module's parameters
  assert rc != 0 and 'carla_garage_amd.LidarCenterNet computes on MI355X only' in out, out[-3000:]  # first forward, on a CPU tensor
