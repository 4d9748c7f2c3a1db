"""One small invocation of the hot path on cuda:0 checked against the CPU oracle (used by __graft_entry__.smoke()).

Lives in tools/, not in the product package: it imports the oracle (the checker), and carla_garage_amd/ is oracle-free by construction
(tests/test_host_helpers.py::test_the_product_never_imports_the_oracle has no exception list)."""
import torch


def run(verbose=True):
  from oracle import tfpp_port as P  # checker only
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  dev = torch.device('cuda:0')
  cfg = GlobalConfig()
  pc = P.PortConfig()
  sd = P.make_state_dict(pc)
  model = LidarCenterNet(cfg)
  model.load_state_dict(sd, strict=True)
  model.to(dev).eval()
  inp = P.make_inputs(1, pc)
  with torch.inference_mode():
    got = model(*[x.to(dev) for x in inp])
    want = P.forward(sd, pc, *inp)
  torch.cuda.synchronize()
  worst = 0.0
  for name, g, w in (('pred_target_speed', got[1], want[1]), ('pred_checkpoint', got[2], want[2]), ('heatmap', got[6][0], want[6][0]),
                     ('pred_semantic', got[3], want[3]), ('pred_bev_semantic', got[4], want[4]), ('pred_depth', got[5], want[5])):
    e = ((g.float().cpu() - w).abs().max() / (w.abs().max() + 1e-20)).item()
    worst = max(worst, e)
    if verbose:
      print(f'smoke {name:20s} rel_err {e:.3e}')
    assert e <= 1e-3, f'{name}: rel err {e:.3e} > 1e-3 vs the CPU oracle'
  # producer of lidar_bev: bit-exact integer/byte work
  import numpy as np
  from oracle import lidar_port as L  # checker only
  from carla_garage_amd.lidar import LidarHistogram
  cloud = L.make_cloud(20000, 5)
  hist = LidarHistogram(cfg, dev)(cloud, True).cpu().numpy()
  assert np.array_equal(hist, L.lidar_to_histogram_features(cloud, True)), 'LiDAR histogram differs from the CPU oracle'
  if verbose:
    print('smoke lidar_histogram      bit-exact')
  # the loader's LiDAR path (section 8(f) item 4): align + histogram of two raw float64 sweeps in one call
  from carla_garage_amd.lidar import LidarBatchHistogram, align_params
  meas = L.make_measurements(3, 2)
  sweeps = [L.make_sweep_f64(5000, 31 + t) for t in range(2)]
  bev = LidarBatchHistogram(cfg, dev)(sweeps, [align_params(meas[t], meas[1], 0.4, -6.0) for t in range(2)], False).cpu().numpy()
  for t in range(2):
    assert np.array_equal(bev[t], L.lidar_to_histogram_features(L.align(sweeps[t], meas[t], meas[1], 0.4, -6.0), False)), 'aligned LiDAR histogram differs from the CPU oracle'
  if verbose:
    print('smoke lidar_align+histogram bit-exact')
  return worst
