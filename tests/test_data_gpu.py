"""DeviceBatchPrefetcher (carla_garage_amd/data.py) against the reference's synchronous upload (team_code/train.py:688-766): same device
tensors (bit-exact: integer / exactly representable conversions), same losses when the train step consumes them, slot reuse under
back-to-back steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_batches(cfg, n, bs=2):
  import bench
  from carla_garage_amd.data import to_reference_batch
  return [to_reference_batch(bench.synthetic_batch(bs, cfg, None, 100 + i), cfg) for i in range(n)]


def _reference_upload(b, cfg, dev):
  """what train.py does, key by key"""
  from carla_garage_amd.data import KEYMAP
  out = {}
  for src, dst, dt, need in KEYMAP:
    if not need(cfg):
      continue
    t = b[src]
    if src == 'route':
      t = t[:, :cfg.predict_checkpoint_len]
    t = t.to(dev, dtype=dt)
    if src == 'speed':
      t = t.unsqueeze(1)
    out[dst] = t
  return out


@pytest.mark.parametrize('src,dst', [(torch.uint8, torch.float32), (torch.uint8, torch.int64), (torch.int32, torch.float32), (torch.int32, torch.int64)])
def test_widen_kernel(src, dst):
  from carla_garage_amd._lib import lib
  from carla_garage_amd.ops import ptr, stream
  kinds = {torch.uint8: 0, torch.int32: 1, torch.float32: 0, torch.int64: 1}
  for n in (0, 1, 3, 4, 15, 16, 17, 4099, 3 * 384 * 1024 * 2):
    x = torch.randint(0, 256, (max(n, 1),), device='cuda').to(src)[:n]
    if src == torch.int32:
      x = x * 65537 - 1000
    y = torch.full((max(n, 1) + 8,), -1, dtype=dst, device='cuda')
    lib.tfpp_widen(ptr(x) if n else ptr(y), ptr(y), n, kinds[src], kinds[dst], stream())
    assert torch.equal(y[:n], x.to(dst))
    assert torch.all(y[n:] == -1)


def test_prefetcher_matches_reference_upload():
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import DeviceBatchPrefetcher
  cfg = GlobalConfig()
  host = _host_batches(cfg, 5)
  got = []
  for b in DeviceBatchPrefetcher(host, cfg):
    got.append({k: v.clone() for k, v in b.items()})  # the slot is recycled two batches later
    # a long-running consumer on the compute stream: the slot must not be overwritten before it has run
    torch.cuda._sleep(20_000_000)
  assert len(got) == len(host)
  for b, h in zip(got, host):
    ref = _reference_upload(h, cfg, 'cuda')
    assert set(ref) == set(b)
    for k in ref:
      assert b[k].dtype == ref[k].dtype and b[k].shape == ref[k].shape, k
      assert torch.equal(b[k], ref[k]), k


def test_slot_not_overwritten_while_consumer_pending():
  """consumer reads the slot late on the compute stream (after a long sleep kernel); contents must still be that batch's"""
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import DeviceBatchPrefetcher
  cfg = GlobalConfig()
  host = _host_batches(cfg, 6)
  sums = []
  for b in DeviceBatchPrefetcher(host, cfg):
    torch.cuda._sleep(50_000_000)
    sums.append(b['rgb'].double().sum() + b['depth_label'].double().sum())
  torch.cuda.synchronize()
  for s, h in zip(sums, host):
    assert float(s) == float(h['rgb'].double().sum() + h['depth'].double().sum())


def test_train_steps_from_prefetcher_equal_resident_batches():
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import DeviceBatchPrefetcher
  from carla_garage_amd.model import LidarCenterNet
  from carla_garage_amd.trainer import Trainer
  cfg = GlobalConfig()
  host = _host_batches(cfg, 3)
  runs = []
  for mode in range(2):
    torch.manual_seed(0)
    m = LidarCenterNet(cfg).cuda().train()
    for mod in m.modules():
      if isinstance(mod, torch.nn.Dropout):
        mod.p = 0.0
    m.config.embd_pdrop = m.config.resid_pdrop = m.config.attn_pdrop = 0.0
    tr = Trainer(m, lr=1e-5)
    it = DeviceBatchPrefetcher(host, cfg) if mode else (_reference_upload(h, cfg, 'cuda') for h in host)
    runs.append(torch.stack([tr.train_step(b).clone() for b in it]).cpu())
  # same inputs -> same losses up to the run-to-run noise of the fp32 atomics in the step (a wrong or stale slot moves them by O(1))
  torch.testing.assert_close(runs[0], runs[1], rtol=1e-2, atol=1e-3)


def test_prefetcher_rasterises_targets_on_device():
  """host batches carry the float64 box lists instead of the nine label maps: same label tensors as the oracle's rasterisation uploaded"""
  import numpy as np
  from oracle import targets_port
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import TARGET_KEYS, DeviceBatchPrefetcher, collate_boxes
  cfg = GlobalConfig()
  bs = 3
  host, want = [], []
  label_src = {'center_heatmap', 'wh', 'offset', 'yaw_class', 'yaw_res', 'velocity', 'brake_target', 'pixel_weight', 'avg_factor'}
  for i, h in enumerate(_host_batches(cfg, 4, bs=bs)):
    lists = [targets_port.make_boxes((5 * i + 11 * j) % 31, 50 + 10 * i + j) for j in range(bs)]
    h = {k: v for k, v in h.items() if k not in label_src}
    h['bounding_boxes_f64'], h['num_bounding_boxes'] = collate_boxes(lists)
    host.append(h)
    want.append([targets_port.get_targets(b) for b in lists])
  names = dict(center_heatmap_target='center_heatmap_label', wh_target='wh_label', offset_target='offset_label', yaw_class_target='yaw_class_label',
               yaw_res_target='yaw_res_label', velocity_target='velocity_label', brake_target='brake_target_label', pixel_weight='pixel_weight_label')
  n = 0
  for b, w in zip(DeviceBatchPrefetcher(host, cfg, rasterise_on_device=True), want):
    assert set(TARGET_KEYS) <= set(b) and 'bounding_boxes_f64' not in b
    got = {k: b[k].cpu().numpy() for k in TARGET_KEYS}
    for j, (t, avg) in enumerate(w):
      for k, mine in names.items():
        if k == 'center_heatmap_target':
          assert np.abs(got[mine][j] - t[k]).max() <= 2.5e-7
        else:
          assert np.array_equal(got[mine][j], t[k].astype(got[mine].dtype)), (k, j)
      assert float(got['avg_factor_label'][j]) == float(avg)
    n += 1
  assert n == 4


@pytest.mark.parametrize('seq', [1, 6])
def test_prefetcher_aligns_and_bins_the_raw_lidar_sweeps_on_device(seq):
  """the host batches carry what CARLA_Data.__getitem__ has BEFORE its numpy LiDAR work -- the raw float64
  sweeps and the ego poses (here through carla_garage_amd.lidar.align_params + data.collate_lidar, per-sample dicts as a Dataset yields them) --
  and the prefetcher produces `lidar_bev` on the copy stream.  Must equal, bit for bit, what the loader would have stacked: the oracle's
  CARLA_Data.align + lidar_to_histogram_features per sample and time frame (pinned on the reference's own output by tests/test_lidar.py),
  concatenated over time as data.py:536,558 concatenates them; the other keys take the usual path."""
  import numpy as np
  from oracle import lidar_port as L
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.data import DeviceBatchPrefetcher, collate_lidar
  from carla_garage_amd.lidar import align_params
  cfg = GlobalConfig(lidar_seq_len=seq) if seq > 1 else GlobalConfig()
  bs, nb = 3, 3
  host, want = [], []
  for i, h in enumerate(_host_batches(cfg, nb, bs=bs)):
    samples, bev = [], []
    if seq > 1:  # the temporal CenterNet labels of a multi-frame configuration (data.py:725-728), absent from the single-frame synthetic batch
      h['velocity'] = torch.zeros(bs, 1, 64, 64)
      h['brake_target'] = torch.zeros(bs, 64, 64, dtype=torch.int32)
    for j in range(bs):
      meas = L.make_measurements(10 * i + j, seq)
      y_aug, yaw_aug = 0.3 * j - 0.2, 4.0 * i - 3.0
      sweeps = [L.make_sweep_f64(3000 + 500 * j + 100 * t, 1000 * i + 10 * j + t) for t in range(seq)]
      if i == 1 and j == 2:
        sweeps[0] = np.zeros((0, 3))  # an empty sweep in the middle of a batch
      samples.append({'lidar_sweeps': sweeps, 'lidar_align': np.stack([align_params(meas[t], meas[seq - 1], y_aug, yaw_aug) for t in range(seq)]),
                      **{k: v[j] for k, v in h.items() if k not in ('lidar', 'temporal_lidar')}})
      bev.append(np.concatenate([L.lidar_to_histogram_features(L.align(sweeps[t], meas[t], meas[seq - 1], y_aug, yaw_aug), cfg.use_ground_plane)
                                 for t in range(seq)], axis=0))
    host.append(collate_lidar(samples, cfg))
    want.append(np.stack(bev))
  n = 0
  for b, w, h in zip(DeviceBatchPrefetcher(host, cfg, lidar_on_device=True), want, host):
    assert b['lidar_bev'].shape == w.shape and b['lidar_bev'].dtype == torch.float32
    assert np.array_equal(b['lidar_bev'].cpu().numpy(), w), f'batch {n}'
    assert torch.equal(b['rgb'], h['rgb'].to('cuda', torch.float32))
    torch.cuda._sleep(10_000_000)
    n += 1
  assert n == nb
