#!/usr/bin/env python
"""bench.py -- TransFuser++ training throughput on MI355X (BASELINE.json metric: training samples/s at bs=12/GPU).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one synthetic batch: forward (train-mode BN, dropout), the 10 losses, the
hand-written backward, the gradient all-reduce (N>1, RCCL) and AdamW(amsgrad) -- carla_garage_amd/trainer.py.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line; it carries
  roofline      the dominant kernel family (largest share of the step's kernel time): algorithmic FLOPs or bytes / HIP-event time on the
                launch stream, against the MFMA peak when its FLOP/byte lies above the ridge (312 FLOP/B for bf16), against HBM otherwise
  roofline_mfma the same object for the dominant MFMA-bound family when that is a different one (the LDS-DMA GEMMs of the fusion transformers)
  roofline_step the whole step: sum of algorithmic FLOPs / GEMM-operand bytes against the measured step, launches, kernel time, and the committed
                rocprofv3 trace of one replayed graph step (profiles/rNN_graph_step.json); every roofline object carries `graph_trace` = its family's
                average launch duration inside that replayed graph
  cpu_baseline  the CPU oracle (oracle/tfpp_port.py, "port") doing the same train step on the host cores, bounded sample
  also: dropin (the step behind train.py's call sites), fp32_step (+ bf16-vs-fp32 gradient cosine), gradient_exchange (N > 1 / --force-collectives),
  fwd_ms_per_frame, lidar_histogram, image_augmentation, video_swin, parity (what is pinned against what)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0  # HBM3E, same guide
GFLOP_PER_SAMPLE_TRAIN = 306.0  # 51.06 GMAC fwd x 2 FLOP x 3 (fwd + bwd)


def synthetic_batch(bs, cfg, device, seed):
  """Synthetic camera + LiDAR batch and labels of SURVEY.md section 8(d) (seeded torch.Generator, seed 1234 + rank)."""
  g = torch.Generator().manual_seed(seed)
  H, W, LH, LW = cfg.camera_height, cfg.camera_width, cfg.lidar_resolution_height, cfg.lidar_resolution_width
  hb, wb = LH // cfg.bev_down_sample_factor, LW // cfg.bev_down_sample_factor
  b = {}
  b['rgb'] = torch.randint(0, 256, (bs, 3, H, W), generator=g).float()
  occ = (torch.rand(bs, 1, LH, LW, generator=g) < 0.1).float()
  b['lidar_bev'] = occ * torch.randint(1, 6, (bs, 1, LH, LW), generator=g).float() / 5.0
  b['target_point'] = torch.randn(bs, 2, generator=g) * torch.tensor([20.0, 5.0])
  b['ego_vel'] = torch.rand(bs, 1, generator=g) * 8.0
  b['command'] = torch.eye(6)[torch.randint(0, 6, (bs,), generator=g)]
  b['target_speed_label'] = torch.randint(0, 4, (bs,), generator=g)
  b['checkpoint_label'] = torch.randn(bs, cfg.predict_checkpoint_len, 2, generator=g) * 5
  b['waypoint_label'] = torch.randn(bs, cfg.pred_len, 2, generator=g) * 5
  b['semantic_label'] = torch.randint(0, cfg.num_semantic_classes, (bs, H, W), generator=g)
  b['bev_semantic_label'] = torch.randint(0, cfg.num_bev_semantic_classes, (bs, LH, LW), generator=g)
  b['depth_label'] = torch.rand(bs, H, W, generator=g)
  heat = torch.zeros(bs, cfg.num_bb_classes, hb, wb)
  pw = torch.zeros(bs, 2, hb, wb)
  ys, xs = torch.meshgrid(torch.arange(hb).float(), torch.arange(wb).float(), indexing='ij')
  nbox = torch.randint(1, 6, (bs,), generator=g)
  for i in range(bs):
    for _ in range(int(nbox[i])):
      y, x, c = [int(torch.randint(lo, hi, (1,), generator=g)) for lo, hi in ((4, hb - 4), (4, wb - 4), (0, cfg.num_bb_classes))]
      blob = torch.exp(-((ys - y)**2 + (xs - x)**2) / (2 * 1.5**2))
      blob[y, x] = 1.0
      heat[i, c] = torch.maximum(heat[i, c], blob)
      pw[i, :, y, x] = 1.0
  b['center_heatmap_label'] = heat
  b['wh_label'] = torch.rand(bs, 2, hb, wb, generator=g) * 8
  b['yaw_class_label'] = torch.randint(0, cfg.num_dir_bins, (bs, hb, wb), generator=g)
  b['yaw_res_label'] = (torch.rand(bs, 1, hb, wb, generator=g) - 0.5) * 0.6
  b['offset_label'] = torch.rand(bs, 2, hb, wb, generator=g)
  b['pixel_weight_label'] = pw
  b['avg_factor_label'] = nbox.float()
  return {k: (v.to(device) if device is not None else v) for k, v in b.items()}


def cpu_baseline_worker(cfg_bs, budget_s):
  """Child process: the oracle ("port": plain-PyTorch fp32 restatement of the reference) on the host cores.  (1) the bs = 1 eval forward of
  BASELINE config 2 (median of 5, the CPU counterpart of fwd_ms_per_frame); (2) the train step of config 3 -- forward + 10 losses + backward
  + AdamW(amsgrad) -- at bs = 12: one warm-up step, then timed steps until three are done or the budget is spent (median).  Prints one JSON
  line after every timed step (the parent keeps the last one, so a slow host still reports what it finished)."""
  from oracle import tfpp_port as P
  # 256-thread hosts thrash on these layer sizes (intra-op parallelism saturates far earlier): cap at 32 threads
  torch.set_num_threads(min(os.cpu_count() or 1, 32))
  pc = P.PortConfig()
  sd = P.make_state_dict(pc)
  inp1 = P.make_inputs(1, pc)
  fwd = []
  with torch.inference_mode():
    for k in range(6):
      t0 = time.perf_counter()
      P.forward(sd, pc, *inp1)
      if k:
        fwd.append(time.perf_counter() - t0)
  fwd_ms = round(1e3 * sorted(fwd)[len(fwd) // 2], 1)
  frozen = lambda k: ('valid_bev' in k or 'running' in k or k.startswith('loss_'))
  sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not frozen(k) else v.clone()) for k, v in sd.items()}
  opt = torch.optim.AdamW([v for v in sd.values() if v.requires_grad], lr=3e-4, amsgrad=True)
  inp = P.make_inputs(cfg_bs, pc)
  lab = P.make_labels(cfg_bs, pc)
  times = []
  t_start = time.perf_counter()
  it = 0
  while True:
    t0 = time.perf_counter()
    out = P.forward(sd, pc, *inp, training=True)
    total, _ = P.total_loss(sd, pc, out, lab)
    opt.zero_grad(set_to_none=True)
    total.backward()
    opt.step()
    dt = time.perf_counter() - t0
    if it > 0 or dt > budget_s / 2:  # the first step is warm-up unless it alone eats the budget
      times.append(dt)
      med = sorted(times)[len(times) // 2]
      print(json.dumps({'value': round(cfg_bs / med, 4), 'unit': 'samples/s', 'cores': torch.get_num_threads(), 'kind': 'port', 'steps': len(times),
                        'fwd_ms_per_frame_bs1': fwd_ms,
                        'sample': f'oracle port (plain-PyTorch fp32 restatement of the reference; /root/reference does not travel to the GPU box): bs={cfg_bs}, '
                                  f'median of {len(times)} train steps (fwd + 10 losses + bwd + AdamW-amsgrad) after one warm-up step: {med:.2f} s per step; '
                                  f'fwd_ms_per_frame_bs1 = median of 5 eval forwards at bs=1; threads capped at 32 of {os.cpu_count()} '
                                  '(intra-op scaling of these layer sizes saturates earlier)'}), flush=True)
    it += 1
    if len(times) >= 3 or (times and time.perf_counter() - t_start + 1.2 * dt > budget_s):
      break


def cpu_baseline(cfg_bs=12, budget_s=75.0, hard_timeout_s=150.0):
  """Runs cpu_baseline_worker in a subprocess with a hard timeout so a slow host can never stall the bench."""
  import subprocess
  cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--batch-size', str(cfg_bs), '--cpu-budget', str(budget_s)]
  env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
  last = None
  try:
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=hard_timeout_s, env=env, check=False)
    out = p.stdout.decode()
  except subprocess.TimeoutExpired as e:
    out = (e.stdout or b'').decode()
  for ln in out.splitlines():
    try:
      last = json.loads(ln)
    except ValueError:
      pass
  if last is None:
    last = {'value': None, 'unit': 'samples/s', 'cores': os.cpu_count(), 'kind': 'port',
            'sample': f'no CPU train step finished within {hard_timeout_s:.0f} s on this host'}
  return last


def bf16_vs_fp32_gradients(batch, device, log, state_dict=None):
  """What the benchmarked precision is: the gradients of ONE bf16 training step against the fp32 HIP step (the reference's arithmetic: use_amp = 0)
  on identical weights, batch and dropout masks -- cosine and relative L2 distance over the whole gradient arena.  (Per-tensor statistics and
  the 200-step loss curves against the autocast reference: tests/test_model.py, profiles/rNN_model_parity_report.jsonl.)"""
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  from carla_garage_amd.trainer import Trainer
  grads = {}
  for dt_ in ('fp32', 'bf16'):
    torch.manual_seed(0)
    m = LidarCenterNet(GlobalConfig(tfpp_dtype=dt_)).to(device).train()
    if state_dict is not None:  # the weights the timed run ended with (at initialisation the last BatchNorm of every block is zero: a shallower network)
      m.load_state_dict(state_dict, strict=True)
    tr = Trainer(m, lr=0.0)
    tr.step_count += 1
    tr.eng.buckets.begin_issue()
    tr._step_body(batch)
    torch.cuda.synchronize()
    grads[dt_] = {n: g.detach().double().flatten().clone() for n, g in tr.eng.grads.items()}
    del tr, m
    torch.cuda.empty_cache()
  from oracle.grad_stats import gradient_stats  # (the checker's formulas: the same ones the autocast reference is measured with)
  st = gradient_stats(grads['fp32'], grads['bf16'])
  out = {k: round(v, 5) for k, v in st.items()}
  out.update(elements=int(sum(g.numel() for g in grads['fp32'].values())), weights='as left by the timed bf16 run' if state_dict is not None else 'initialisation')
  log(f'bf16 vs fp32 gradients of one step: {out}')
  return out


def autocast_reference_worker(sd_path, bs, seed):
  """Child process (CPU): the bf16 REFERENCE for the weights a bench run ends with -- the oracle port under torch.autocast('cpu', bfloat16)
  (what team_code/train.py:885's autocast does to the reference; tests/test_oracle.py pins port-under-autocast on the reference-under-autocast
  fixture) against the port's own fp32 step, same weights, same batch, same dropout masks; statistics of oracle/grad_stats.py."""
  from oracle import tfpp_port as P
  from oracle.grad_stats import gradient_stats
  from carla_garage_amd.config import GlobalConfig
  torch.set_num_threads(min(os.cpu_count() or 1, 32))
  pc = P.PortConfig()
  weights = torch.load(sd_path) if sd_path else P.make_state_dict(pc)
  b = synthetic_batch(bs, GlobalConfig(), None, seed)
  inp = [b[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
  frozen = lambda k: ('valid_bev' in k or 'running' in k or 'num_batches' in k or k.startswith('loss_'))
  grads, losses = {}, {}
  t0 = time.perf_counter()
  for mode in ('fp32', 'bf16'):
    sd = {k: (v.detach().clone().float().requires_grad_(True) if v.is_floating_point() and not frozen(k) else v.detach().clone()) for k, v in weights.items()}
    torch.manual_seed(1)  # the same dropout masks in both runs
    with torch.autocast('cpu', dtype=torch.bfloat16, enabled=mode == 'bf16'):
      out = P.forward(sd, pc, *inp, training=True)
      total, ls = P.total_loss(sd, pc, out, b)
    total.float().backward()
    grads[mode] = {k: v.grad.detach().float() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    losses[mode] = float(total)
  st = gradient_stats(grads['fp32'], grads['bf16'])
  print(json.dumps({**{k: round(v, 5) for k, v in st.items()}, 'weighted_loss_fp32': round(losses['fp32'], 5), 'weighted_loss_autocast': round(losses['bf16'], 5),
                    'cpu_seconds': round(time.perf_counter() - t0, 1), 'threads': torch.get_num_threads()}), flush=True)


def autocast_reference(state_dict, bs, seed, log, hard_timeout_s=240.0):
  """Runs autocast_reference_worker in a subprocess (no GPU visible); returns its statistics or {'error': ...}."""
  import subprocess
  import tempfile
  path = ''
  try:
    if state_dict is not None:
      fd, path = tempfile.mkstemp(suffix='.pt', dir='/tmp')
      os.close(fd)
      torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, path)
    cmd = [sys.executable, os.path.abspath(__file__), '--autocast-reference-only', path, '--batch-size', str(bs), '--seed', str(seed)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=hard_timeout_s, env=env, check=False)
    for ln in reversed(p.stdout.decode().splitlines()):
      try:
        return json.loads(ln)
      except ValueError:
        continue
    return {'error': 'no result', 'stderr': p.stderr.decode()[-300:]}
  except Exception as e:  # pylint: disable=broad-except
    log(f'autocast reference leg failed: {type(e).__name__}: {e}')
    return {'error': f'{type(e).__name__}: {e}'}
  finally:
    if path and os.path.exists(path):
      os.remove(path)


def forward_call_worker(device_index, iters=20):
  """Child process (an agent-like one: nothing but the module and its eval forward): model.forward() exactly as sensor_agent.py:456-461 calls it, no
  switch in the environment -- the module captures the eval forward of a signature after two eager calls and replays it (model.py
  _plain_forward).  `_tick` = each call followed by a device synchronisation (the agent reads the predictions on the host before the next tick)."""
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  device = torch.device('cuda', device_index)
  torch.cuda.set_device(device)
  cfg = GlobalConfig(tfpp_dtype='bf16')
  torch.manual_seed(0)
  model = LidarCenterNet(cfg).to(device).eval()
  b = synthetic_batch(1, cfg, device, 99)
  inp = [b[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
  out = {}
  for dtype in ('bf16', 'fp32'):
    cfg.tfpp_dtype = dtype
    with torch.inference_mode():
      for _ in range(5):  # two eager calls, the capture, two replays
        model(*inp)
      for key, sync_each in ((f'{dtype}_forward_call', False), (f'{dtype}_forward_call_tick', True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
          model(*inp)
          if sync_each:
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        out[key] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
  out['captured_signatures'] = sum(pl.get('graph') is not None for pl in model._eval_plans.values())
  print(json.dumps(out), flush=True)


def forward_call_latency(device, log, hard_timeout_s=150.0):
  """forward_call_worker in its own process on the same GPU: whatever happens there cannot take this process (and the bench line) down."""
  import subprocess
  try:
    cmd = [sys.executable, os.path.abspath(__file__), '--forward-call-only', str(device.index or 0)]
    env = dict(os.environ)  # (no switch: the module's own capture of repeated eval calls is the default since round 6)
    env.pop('TFPP_EVAL_GRAPH_AFTER', None)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
      env.pop(k, None)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=hard_timeout_s, env=env, check=False)
    for ln in reversed(p.stdout.decode().splitlines()):
      try:
        return json.loads(ln)
      except ValueError:
        continue
    log(f'forward-call leg: no result (rc {p.returncode}): {p.stderr.decode()[-300:]}')
  except Exception as e:  # pylint: disable=broad-except
    log(f'forward-call leg failed: {type(e).__name__}: {e}')
  return {}


def inference_latency(model, cfg, device, log, iters=20):
  """Second half of the BASELINE metric: TransFuser++ forward ms/frame at bs=1 (the 20 Hz closed-loop tick,
  sensor_agent.py:456-461), eval mode, caller-facing fp32 NCHW outputs included; eager launches and hipGraph replay."""
  from carla_garage_amd.graph import GraphedForward
  b = synthetic_batch(1, cfg, device, 99)
  inp = [b[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
  out = {}
  model.eval()
  def timed(fn, sync_each=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      fn()
      if sync_each:
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / iters, 3)

  for dtype in ('bf16', 'fp32'):
    cfg.tfpp_dtype = dtype
    with torch.inference_mode():
      model.eval_graph_after = -1  # launches issued one by one
      for _ in range(3):
        model(*inp)
      out[f'{dtype}_eager'] = timed(lambda: model(*inp))
    try:
      g = GraphedForward(model, *inp)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(iters):
        g(*inp)
      torch.cuda.synchronize()
      out[f'{dtype}_hipgraph'] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
    except Exception as e:  # pylint: disable=broad-except
      out[f'{dtype}_hipgraph'] = None
      log(f'inference hipGraph capture failed: {type(e).__name__}: {e}')
    log(f'forward bs=1 {dtype}: {out}')
  model.train()
  return out


def video_swin_forward(device, log, bs=4, iters=10, train=True, roofline=True):
  """BASELINE config 5 (TransFuser++ with the Video-Swin LiDAR branch, 6 LiDAR frames -> 3 time frames per scale, bs = 4 per GPU):
  inference forward and training step in bf16, random-init weights, synthetic frames, as hipGraph replays.  Reported beside the headline
  metric (BASELINE config 3), never as ``value``."""
  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.graph import GraphedForward
  from carla_garage_amd.model import LidarCenterNet
  cfg = GlobalConfig(lidar_architecture='video_swin_tiny', lidar_seq_len=6, tfpp_dtype='bf16')
  torch.manual_seed(0)
  model = LidarCenterNet(cfg).to(device).eval()
  b = synthetic_batch(bs, cfg, device, 77)
  g = torch.Generator().manual_seed(78)
  occ = (torch.rand(bs, 6, cfg.lidar_resolution_height, cfg.lidar_resolution_width, generator=g) < 0.1).float()
  b['lidar_bev'] = (occ * torch.randint(1, 6, occ.shape, generator=g).float() / 5.0).to(device)
  inp = [b[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')]
  out = {'batch': bs, 'dtype': 'bf16', 'lidar_frames': 6}
  model.eval_graph_after = -1  # (the eager number; the replay is measured through GraphedForward below)
  with torch.inference_mode():
    for _ in range(3):
      model(*inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      model(*inp)
    torch.cuda.synchronize()
    out['eager_ms_per_batch'] = round(1e3 * (time.perf_counter() - t0) / iters, 3)
  try:
    gf = GraphedForward(model, *inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      gf(*inp)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    out['hipgraph_ms_per_batch'] = round(ms, 3)
    out['hipgraph_samples_per_s'] = round(bs / (ms * 1e-3), 1)
  except Exception as e:  # pylint: disable=broad-except
    out['hipgraph_ms_per_batch'] = None
    log(f'video-swin hipGraph capture failed: {type(e).__name__}: {e}')
  log(f'video-swin forward bs={bs}: {out}')
  # the same configuration trained: fwd + 12 losses + bwd + AdamW(amsgrad), train-mode BN, dropout and stochastic depth on
  # (single-process runs only: a Trainer broadcasts / all-reduces over the default group, and only rank 0 executes this leg)
  try:
    if not train:
      raise RuntimeError('skipped in multi-rank runs')
    from carla_garage_amd.graph import GraphedTrainStep
    from carla_garage_amd.trainer import Trainer
    model.train()
    tb = dict(b)
    hb, wb = cfg.lidar_resolution_height // cfg.bev_down_sample_factor, cfg.lidar_resolution_width // cfg.bev_down_sample_factor
    tb['velocity_label'] = torch.rand(bs, 1, hb, wb, generator=g).to(device) * 8.0
    tb['brake_target_label'] = torch.randint(0, 2, (bs, hb, wb), generator=g).to(device)
    tr = Trainer(model, lr=cfg.lr)
    tr.train_step(tb)
    step = GraphedTrainStep(tr, tb)
    for _ in range(2):
      step(tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
      vals = step(tb)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / iters
    out['train_hipgraph_ms_per_step'] = round(ms, 3)
    out['train_samples_per_s'] = round(bs / (ms * 1e-3), 1)
    out['train_final_weighted_loss'] = round(float(tr.total_loss(vals)), 5)
    try:  # roofline of this configuration's step: algorithmic FLOPs / GEMM-family bytes from two profiled eager steps over the replayed step time
      if not roofline:
        raise RuntimeError('skipped (kernel-trace run: the last step of the trace must be a replayed one)')
      from carla_garage_amd._lib import lib, KernelProfiler
      prof = KernelProfiler()
      lib.profiler = prof
      for _ in range(2):
        tr.step_count += 1
        tr.eng.buckets.begin_issue()
        tr._step_body(tb)
        tr._optimizer(tr.step_count)
      lib.profiler = None
      agg = prof.summary()
      flop = sum(x['flops'] for x in agg.values()) / 2
      gb = sum(x['bytes'] for x in agg.values()) / 2
      peak = PEAK_TFLOPS['bf16']
      fam, a = max(((f, x) for f, x in agg.items() if x['flops'] > 0), key=lambda fa: fa[1]['ms'])
      roof5 = {'workload': 'TransFuser++ with the Video-Swin LiDAR branch (BASELINE config 5), training step bs = 4 bf16, 6 LiDAR frames',
               'algorithmic_tflop_per_step': round(flop / 1e12, 4), 'algorithmic_gb_gemm_families': round(gb / 1e9, 3), 'ms_per_step': round(ms, 3),
               'achieved_tflops': round(flop / (ms * 1e-3) / 1e12, 2), 'frac_of_mfma_peak': round(flop / (ms * 1e-3) / 1e12 / peak, 4),
               'at_mfma_peak_ms': round(1e3 * flop / (peak * 1e12), 3), 'at_hbm_peak_ms_gemm_families': round(1e3 * gb / (PEAK_HBM_GBS * 1e9), 3),
               'launches_per_step_library_calls': sum(x['calls'] for x in agg.values()) // 2,
               'dominant_family': {'kernel': fam, 'launches_per_step': a['calls'] // 2, 'avg_launch_us': round(1e3 * a['ms'] / a['calls'], 2),
                                   'achieved_tflops': round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 2), 'frac_of_mfma_peak': round(a['flops'] / (a['ms'] * 1e-3) / 1e12 / peak, 4),
                                   'share_of_step_kernel_time': round(a['ms'] / sum(x['ms'] for x in agg.values()), 3)},
               'families_by_time': [{'kernel': f, 'launches_per_step': x['calls'] // 2, 'ms_per_step': round(x['ms'] / 2, 3)}
                                    for f, x in sorted(agg.items(), key=lambda fa: -fa[1]['ms'])[:8]]}
      import glob
      files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_swin_graph_step.json')))
      if files:
        with open(files[-1], encoding='utf-8') as f:
          g5 = json.load(f)
        roof5['graph_trace'] = {'source': 'profiles/' + os.path.basename(files[-1]), 'launches': g5['launches'], 'span_ms': g5['span_ms'],
                                'sum_kernel_ms': g5['sum_kernel_ms'], 'busy_union_ms': g5['busy_union_ms']}
      out['roofline_config5'] = roof5
    except Exception as e:  # pylint: disable=broad-except
      try:
        from carla_garage_amd._lib import lib as _l
        _l.profiler = None
      except Exception:  # pylint: disable=broad-except
        pass
      log(f'video-swin roofline: {type(e).__name__}: {e}')
    del step, tr
  except Exception as e:  # pylint: disable=broad-except
    out['train_hipgraph_ms_per_step'] = None
    import traceback
    log(f'video-swin training step failed: {type(e).__name__}: {e}\n' + ''.join(traceback.format_exc().splitlines(True)[-12:]))
  log(f'video-swin bs={bs}: {out}')
  del model
  torch.cuda.empty_cache()
  return out


def lidar_histogram_latency(cfg, device, log, n=60000, iters=50):
  """the LiDAR -> BEV histogram that feeds forward() on every tick (data.py:873-906).  HIP path with
  the points already on the device, and the CPU oracle (numpy, 1 core) beside it as the reported baseline."""
  import numpy as np
  from carla_garage_amd.lidar import LidarHistogram
  from oracle import lidar_port
  cloud = lidar_port.make_cloud(n, 11)
  pts = torch.from_numpy(cloud).to(device)
  hist = LidarHistogram(cfg, device)
  res = hist(pts, False)
  assert np.array_equal(res.cpu().numpy(), lidar_port.lidar_to_histogram_features(cloud, False)), 'LiDAR histogram differs from the oracle'
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    hist(pts, False, out=res)
  e1.record()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5):
    lidar_port.lidar_to_histogram_features(cloud, False)
  cpu = (time.perf_counter() - t0) / 5
  r = {'hip': round(1e3 * e0.elapsed_time(e1) / iters, 2), 'cpu_numpy_1core': round(1e6 * cpu, 1), 'bit_exact': True}
  log(f'LiDAR histogram ({n} points): {r}')
  return r


def image_augmentation_latency(cfg, device, log, bs=12, iters=20):
  """the colour augmentation of the loader (team_code/data.py:1141-1157, color_aug_prob = 0.5) on the uploaded
  uint8 frames of one batch -- host sampling of the programs + the device stages (carla_garage_amd/augment.py), per batch of `bs` frames."""
  from carla_garage_amd.augment import ImageAugmenter
  aug = ImageAugmenter(prob=cfg.color_aug_prob if hasattr(cfg, 'color_aug_prob') else 0.5, seed=1)
  rgb = torch.randint(0, 256, (bs, 3, cfg.camera_height if hasattr(cfg, 'camera_height') else 256, cfg.camera_width if hasattr(cfg, 'camera_width') else 1024),
                      device=device, dtype=torch.uint8)
  aug.apply(rgb)
  torch.cuda.synchronize()
  stages = 0
  t0 = time.perf_counter()
  for _ in range(iters):
    aug.apply(rgb)
    stages += int((aug.last_programs['kind'] != 0).sum(1).max())
  torch.cuda.synchronize()
  r = {'batch': bs, 'us_per_batch': round(1e6 * (time.perf_counter() - t0) / iters, 1), 'avg_stages': round(stages / iters, 2),
       'note': 'host program sampling + device stages, frames resident; the reference runs imgaug per sample in the DataLoader workers'}
  log(f'image augmentation: {r}')
  return r


def dropin_step_time(model, cfg, batch, steps, warmup, optimizer, log, ddp=False):
  """ms/step of the DROP-IN boundary: the module driven exactly as team_code/train.py:776-910 drives the reference's (forward with keyword
  arguments -> model.compute_loss -> weighted sum with a host read of every loss (train.py:896) -> backward -> optimizer.step ->
  zero_grad(set_to_none=True)), inside DistributedDataParallel when a process group exists (train.py:516-520).  ``optimizer``: 'fused' =
  carla_garage_amd.optim.FlatAdamW (the one-line substitution of INTEGRATION.md), 'torch' = the unmodified torch.optim.AdamW(amsgrad=True)
  of train.py:529-531.  After TFPP_DROPIN_GRAPH_AFTER eager steps the three phases replay as hipGraphs (carla_garage_amd/dropin.py)."""
  from carla_garage_amd.losses import normalized_loss_weights
  from carla_garage_amd.optim import FlatAdamW
  net = model
  if ddp:
    net = torch.nn.parallel.DistributedDataParallel(model, device_ids=None, output_device=None, broadcast_buffers=False, find_unused_parameters=False)
  opt = (FlatAdamW if optimizer == 'fused' else torch.optim.AdamW)(net.parameters(), lr=cfg.lr, amsgrad=True)
  w = normalized_loss_weights(cfg)
  inp = {k: batch[k] for k in ('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command')}
  lab = {k: v for k, v in batch.items() if k.endswith('_label')}
  lab.setdefault('velocity_label', None)
  lab.setdefault('brake_target_label', None)
  dev = batch['rgb'].device

  def step():
    pred = net(**inp)
    losses = model.compute_loss(pred_wp=pred[0], pred_target_speed=pred[1], pred_checkpoint=pred[2], pred_semantic=pred[3], pred_bev_semantic=pred[4],
                                pred_depth=pred[5], pred_bounding_box=pred[6], pred_wp_1=pred[8], selected_path=pred[9], **lab)
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    detailed = 0.0
    for key, value in losses.items():
      loss += w[key] * value
      detailed += float(w[key] * float(value.item()))  # train.py:896: one host sync per loss
    loss.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    return float(loss.item())  # train.py:913

  opt.zero_grad(set_to_none=False)
  for _ in range(max(warmup, 4)):  # (>= TFPP_DROPIN_GRAPH_AFTER + 2: the timed steps are hipGraph replays)
    last = step()
  torch.cuda.synchronize()
  if ddp:
    dist.barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    last = step()
  torch.cuda.synchronize()
  if ddp:
    dist.barrier()
  el = time.perf_counter() - t0
  if ddp:
    tm = torch.tensor([el], device=dev, dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    el = float(tm.item())
  plan = next(iter(model.__dict__['_dropin_step'].plans.values()))
  r = {'ms_per_step': round(1e3 * el / steps, 3), 'final_weighted_loss': round(last, 5), 'hipgraph': plan.B1 is not None, 'optimizer': optimizer, 'ddp': bool(ddp)}
  log(f'drop-in step ({optimizer} optimizer{", DDP" if ddp else ""}): {r}')
  del opt, net
  return r


XGMI_LINK_GBS = 153.0  # per direction and link; 7 links per GPU, one to every other GPU of the node (MI355X_MICROARCH.md)


def _flush_c_stdio():
  """fflush(NULL): output written through C stdio by the native libraries (RCCL's version banner) leaves the process now."""
  try:
    import ctypes
    ctypes.CDLL(None).fflush(None)
  except Exception:  # pylint: disable=broad-except
    pass


def graph_trace():
  """The newest committed rocprofv3 kernel trace of ONE hipGraph-replayed step (profiles/rNN_graph_step.json, tools/graph_step_profile.sh): launch
  durations as they are inside the replayed graph, next to co-running lanes -- the HIP-event times of the eager profiling steps below are ~10 %
  longer and blind to that.  A committed measurement of the same command, not a live one: `source` names the file."""
  import glob
  files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r[0-9][0-9]_graph_step.json')))
  if not files:
    return None
  try:
    with open(files[-1], encoding='utf-8') as f:
      g = json.load(f)
    g['source'] = 'profiles/' + os.path.basename(files[-1])
    return g
  except (OSError, ValueError):
    return None


def kernel_pattern(fam):
  """Substring of the demangled kernel name that identifies the kernels of family `fam` in a rocprofv3 table (None: no single kernel)."""
  import re
  if fam == 'conv_wgrad<bf16,group256>':  # the grouped weight-gradient grids of a lane batch, one family per member kernel (ops.wgrad_batch_end)
    return 'conv_wgrad_glds_group_kernel<256, 256,'
  if fam == 'conv_wgrad<bf16,group128>':
    return 'conv_wgrad_glds_group_kernel<128, 128,'
  if fam == 'conv_wgrad<bf16,group64>':
    return 'conv_wgrad_glds_group_kernel<64, 64,'
  if fam.startswith('conv_wgrad<bf16,halo3x3>'):
    return 'wgrad3x3_halo_kernel<'
  mh = re.match(r'conv_gemm<bf16,halo8x32x(\d+)>', fam)
  if mh:
    return f'conv3x3_halo_kernel<{int(mh.group(1)) // 16},'
  m = re.match(r'(conv_gemm|conv_wgrad)<(f32|bf16),(glds|halo|lds)?(\d+)x(\d+)', fam)
  if m is None:
    return None
  op, dtype, kind, a, b = m.groups()
  ty = 'float' if dtype == 'f32' else 'unsigned short'
  if kind == 'glds':
    return f'{op}_glds_kernel<{a}, {b},'
  if kind is None or kind == 'lds':
    return f'{op}_kernel<{ty}, {a}, {b},'
  return None


def mfma_counters_of(fam):
  """MFMA-pipe occupancy of family `fam` from the committed counter pass (profiles/pmc_mfma.json, tools/pmc_mfma.sh: rocprofv3 --pmc MfmaUtil =
  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs), and SQ_INSTS_VALU_MFMA_MOPS_* x 512 = executed matrix FLOPs), kernels alone on the chip."""
  pat = kernel_pattern(fam)
  try:
    with open(os.path.join(ROOT, 'profiles', 'pmc_mfma.json'), encoding='utf-8') as f:
      db = json.load(f)
  except (OSError, ValueError):
    return None
  if pat is None:
    return None
  hit = [v for k, v in db.get('kernels', {}).items() if pat.replace('(anonymous namespace)::', '') in k]
  if not hit:
    return None
  n = sum(v['dispatches'] for v in hit)
  util = [v['MfmaUtil_pct'] * v['dispatches'] for v in hit if v.get('MfmaUtil_pct') is not None]
  return {'source': 'profiles/pmc_mfma.json', 'MfmaUtil_pct': round(sum(util) / n, 2) if util else None,
          'executed_mfma_gflop_per_launch': round(sum(v['mfma_flops_bf16'] + v['mfma_flops_f32'] for v in hit) / n / 1e9, 3),
          'note': 'rocprofv3 --pmc, kernel alone on the chip (counters serialise the launches); MfmaUtil = MFMA-busy cycles / (GPU-active cycles x SIMDs)'}


def graph_trace_of(gtrace, fam, flop_per_launch, bytes_per_launch, mfma, peak):
  """Average launch duration of kernel family `fam` inside the replayed graph (graph_trace()), with the roofline fraction it gives."""
  if gtrace is None:
    return None
  pat = kernel_pattern(fam)
  if pat is None:
    return None
  hit = [v for k, v in gtrace['kernels'].items() if pat in k]
  if not hit:
    return None
  calls, total = sum(v['calls'] for v in hit), sum(v['total_ms'] for v in hit)
  us = 1e3 * total / calls
  ach = (flop_per_launch / (us * 1e-6) / 1e12) if mfma else (bytes_per_launch / (us * 1e-6) / 1e9)
  return {'source': gtrace['source'], 'kernel_name_contains': pat, 'launches_per_step': calls, 'avg_launch_us': round(us, 2), 'achieved': round(ach, 2),
          'frac': round(ach / peak, 4)}


def exchange_model(bucket_bytes, world):
  """What the gradient exchange should cost on one 8 x MI355X node, to check the first real multi-GPU run against: S bytes all-reduced over N
  ranks move 2 S (N-1)/N bytes out of every GPU; a direct (one-shot reduce-scatter + all-gather over the full xGMI mesh) algorithm spreads them
  over the N-1 links, a ring is bound by one link.  The arena is exchanged bucket by bucket in the order backward completes them
  (carla_garage_amd/buckets.py), each all-reduce starting behind its bucket's completion signal while the rest of backward runs: what cannot
  hide is the LAST bucket (stage 1 + stems: the smallest parameters of the network), which is complete only when backward ends."""
  n = max(world, 2)
  per_gpu = lambda b: 2.0 * b * (n - 1) / n
  ms = lambda b, links: round(per_gpu(b) / (links * XGMI_LINK_GBS * 1e9) * 1e3, 3)
  total = sum(bucket_bytes)
  return {'ranks_modelled': n, 'arena_bytes': total, 'bucket_bytes': list(bucket_bytes),
          'whole_arena_ms': {'direct': ms(total, n - 1), 'ring': ms(total, 1)},
          'exposed_last_bucket_ms': {'direct': ms(bucket_bytes[-1], n - 1), 'ring': ms(bucket_bytes[-1], 1)},
          'note': f'{XGMI_LINK_GBS:.0f} GB/s per xGMI link and direction, {n - 1} links used by the direct algorithm; one captured graph, no join of the '
                  'lanes before the end of backward'}


def pmc_traffic(family):
  """HBM-side bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written
  by tools/pmc_traffic.sh; bench.py cannot collect PMC counters itself).  An entry records the sha of the kernel source it was
  measured on: after the kernel changes the entry is refused (traffic = null) instead of silently going stale."""
  import hashlib
  try:
    with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), encoding='utf-8') as f:
      ent = json.load(f).get(family)
  except (OSError, ValueError):
    ent = None
  if not ent:
    # no per-family pass: the per-kernel table of the whole step (profiles/pmc_all_kernels.json, tools/pmc_all.sh) by kernel name
    pat = kernel_pattern(family)
    try:
      with open(os.path.join(ROOT, 'profiles', 'pmc_all_kernels.json'), encoding='utf-8') as f:
        allk = json.load(f)
    except (OSError, ValueError):
      allk = None
    if pat and allk:
      hit = [v for k, v in allk['kernels'].items() if pat in k]
      n = sum(v['launches'] for v in hit)
      if n:
        tot = sum(v['fetch_bytes'] + v['write_bytes'] for v in hit)
        return int(tot / n), (f"mean over {n} launches of the kernels matching '{pat}' in profiles/pmc_all_kernels.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, "
                              f"whole-step counter passes of {allk.get('measured', '?')}; not tied to a source hash)")
    return None, 'no PMC measurement committed for this kernel family'
  try:
    with open(os.path.join(ROOT, 'carla_garage_amd', 'csrc', ent['source_file']), 'rb') as f:
      sha = hashlib.sha256(f.read()).hexdigest()[:16]
  except OSError:
    sha = None
  if sha != ent.get('source_sha16'):
    return None, f"stale: {ent['source_file']} changed since the PMC pass of {ent.get('measured', '?')} (re-run tools/pmc_traffic.sh)"
  traffic = ent['fetch_bytes_per_launch'] + (ent.get('write_bytes_per_launch') or 0)
  note = ('FETCH_SIZE x2 (gfx950 correction)' + (' + WRITE_SIZE x1024 B (calibrated on fill kernels, profiles/r02_pmc_calibration.txt)' if ent.get('write_bytes_per_launch') else '') +
          f", separate counter-only rocprofv3 --pmc passes, {ent.get('measured', '')}, profiles/pmc_traffic.json")
  return traffic, note


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--batch-size', type=int, default=12)
  ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-roofline', action='store_true')
  ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay of the step body')
  ap.add_argument('--no-inference', action='store_true', help='skip the bs=1 forward latency measurement')
  ap.add_argument('--no-dropin', action='store_true', help='skip the drop-in boundary leg (module driven as team_code/train.py drives it)')
  ap.add_argument('--kernel-table', action='store_true', help='print the per-kernel-family time table to stderr')
  ap.add_argument('--force-collectives', action='store_true',
                  help='initialise a process group and issue the gradient all-reduces even with one rank (exercises the RCCL path on one GPU)')
  ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
  ap.add_argument('--autocast-reference-only', default=None, help=argparse.SUPPRESS)
  ap.add_argument('--forward-call-only', default=None, help=argparse.SUPPRESS)
  ap.add_argument('--config5-only', action='store_true', help='only the Video-Swin configuration (BASELINE config 5): training step bs = 4 bf16 + its roofline; '
                                                             'what tools/graph_step_profile.sh swin traces')
  ap.add_argument('--seed', type=int, default=1234, help=argparse.SUPPRESS)
  ap.add_argument('--no-bf16-reference', action='store_true', help='skip the CPU autocast-bf16 reference beside the bf16-vs-fp32 gradient statistics')
  ap.add_argument('--cpu-budget', type=float, default=25.0, help=argparse.SUPPRESS)
  args = ap.parse_args()
  if args.autocast_reference_only is not None:
    autocast_reference_worker(args.autocast_reference_only, args.batch_size, args.seed)
    return
  if args.cpu_baseline_only:
    cpu_baseline_worker(args.batch_size, args.cpu_budget)
    return
  if args.forward_call_only is not None:
    forward_call_worker(int(args.forward_call_only))
    return

  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    # started by hand as `python bench.py --gpus N`: become the launcher the driver uses (team_code/shell_train.sh:12 is the
    # reference's torchrun line) -- one rank per GPU, RCCL rendezvous on 127.0.0.1 -- instead of quietly measuring one GPU
    if torch.cuda.device_count() < args.gpus:
      raise SystemExit(f'bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node; refusing to report a '
                       f'{args.gpus}-GPU number from fewer devices')
    import socket
    import subprocess
    with socket.socket() as sk:
      sk.bind(('127.0.0.1', 0))
      port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if args.gpus != world:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python -m torch.distributed.run '
                     f'--nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)')
  if local_rank >= torch.cuda.device_count():
    raise SystemExit(f'rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} GPU(s) visible')
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  rccl_ranks = None
  if world > 1 or args.force_collectives:
    # one process per GPU, RCCL over xGMI (backend 'nccl' is RCCL on ROCm).  --force-collectives builds the group even for one
    # rank, so a 1-GPU box executes the whole exchange path (one graph, in-graph completion signals, one all-reduce per gradient bucket)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    if args.force_collectives:
      os.environ['TFPP_FORCE_COLLECTIVES'] = '1'
    dist.init_process_group('nccl', init_method='env://', rank=rank, world_size=world)
    probe = torch.ones(1, device=device)
    dist.all_reduce(probe)  # a real collective: the reported rank count is what RCCL summed, not what the environment claims
    torch.cuda.synchronize()
    rccl_ranks = int(probe.item())
    assert rccl_ranks == world, f'RCCL all-reduce over {rccl_ranks} ranks, expected {world}'

  if args.config5_only:
    if world != 1:
      raise SystemExit('--config5-only is a single-GPU leg')
    t_c5 = time.perf_counter()
    res = video_swin_forward(device, lambda m: print(f'[bench +{time.perf_counter() - t_c5:7.1f}s] {m}', file=sys.stderr, flush=True), iters=max(4, args.steps), train=True, roofline=os.environ.get('TFPP_CONFIG5_NO_ROOFLINE', '0') != '1')
    torch.cuda.synchronize()
    print(json.dumps({'metric': 'training samples/sec (TransFuser++ Video-Swin, bs=4/GPU: BASELINE config 5; not the headline metric)',
                      'value': res.get('train_samples_per_s'), 'unit': 'samples/s', 'n_gpus': 1, 'dtype': 'bf16', 'data': 'synthetic', 'video_swin_forward_bs4': res}), flush=True)
    return

  from carla_garage_amd.config import GlobalConfig
  from carla_garage_amd.model import LidarCenterNet
  from carla_garage_amd.trainer import Trainer
  from carla_garage_amd._lib import lib, KernelProfiler

  torch.manual_seed(0)  # identical random-init weights on every rank
  cfg = GlobalConfig(tfpp_dtype=args.dtype)
  model = LidarCenterNet(cfg)
  for m in model.modules():  # zero_init_last leaves residual branches inert; give BN gains non-trivial values
    if isinstance(m, torch.nn.BatchNorm2d):
      torch.nn.init.uniform_(m.weight, 0.5, 1.0)
  model.to(device).train()
  t_wall = time.perf_counter()

  def log(msg):
    if rank == 0:
      print(f'[bench +{time.perf_counter() - t_wall:7.1f}s] {msg}', file=sys.stderr, flush=True)

  trainer = Trainer(model, lr=cfg.lr)
  batch = synthetic_batch(args.batch_size, cfg, device, 1234 + rank)
  log('model, trainer and synthetic batch ready')

  def sync():
    if rccl_ranks is not None:
      dist.barrier()
    torch.cuda.synchronize()

  step_fn = lambda: trainer.train_step(batch)
  graphed = False
  if not args.no_graph:
    try:  # capture the step body (repack + fwd + losses + bwd) into one hipGraph; all-reduce + optimizer stay eager
      from carla_garage_amd.graph import GraphedTrainStep
      gstep = GraphedTrainStep(trainer, batch, warmup=1)
      step_fn = lambda: gstep()
      graphed = True
      log('training step captured into a hipGraph')
    except Exception as e:  # pylint: disable=broad-except
      log(f'hipGraph capture unavailable ({type(e).__name__}: {e}); running the eager launch sequence')
  for _ in range(args.warmup):
    vals = step_fn()
  sync()
  log(f'{args.warmup} warm-up steps done')
  t0 = time.perf_counter()
  for _ in range(args.steps):
    vals = step_fn()
  sync()
  elapsed = time.perf_counter() - t0
  log(f'{args.steps} timed steps: {1e3 * elapsed / args.steps:.1f} ms/step')
  tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
  if rccl_ranks is not None:
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
  elapsed = float(tmax.item())
  loss_total = trainer.total_loss(vals)
  assert loss_total == loss_total and abs(loss_total) < 1e9, f'training diverged: {loss_total}'

  comm = None
  if rccl_ranks is not None:
    # exposed communication = step time with the gradient exchange - time of the identical local step (no collective; every rank
    # runs it, so nobody waits in one).  The local step is captured the same way as the real one (single-segment hipGraph).
    trainer.exchange = False
    try:
      local_fn = lambda: trainer.train_step(batch)
      if graphed:
        from carla_garage_amd.graph import GraphedTrainStep
        glocal = GraphedTrainStep(trainer, batch, warmup=1)
        local_fn = lambda: glocal()
      for _ in range(max(1, args.warmup // 2)):
        local_fn()
      sync()
      t1 = time.perf_counter()
      for _ in range(args.steps):
        local_fn()
      sync()
      tl = torch.tensor([time.perf_counter() - t1], device=device, dtype=torch.float64)
      if world > 1:
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
      local_ms = 1e3 * float(tl.item()) / args.steps
      comm = {'rccl_ranks': rccl_ranks, 'ms_per_step_without_exchange': round(local_ms, 3),
              'exposed_comm_ms_per_step': round(1e3 * elapsed / args.steps - local_ms, 3),
              'allreduce_bytes_per_step': int(trainer.eng.flat_grad.numel()) * 4,
              'overlap': 'one hipGraph; one all-reduce per gradient bucket (completion order of backward), each behind its own in-graph completion signal',
              'buckets': len(trainer.eng.buckets.ranges()), 'early_signals_per_step': len(gstep.program[1]) if graphed else len(trainer.program[1]),
              'signal_wait_timeouts': trainer.eng.buckets.timed_out(),
              'predicted': exchange_model([4 * (hi - lo) for lo, hi in trainer.eng.buckets.ranges()], world)}
      log(f'gradient exchange: {comm}')
    finally:
      trainer.exchange = True

  roof = roof_mfma = roof_hbm = roof_fusion = roof_step = None
  gtrace = graph_trace()
  if rank == 0 and rccl_ranks is None and not args.no_roofline:  # per-kernel timing runs extra local steps: single-process runs only
    prof = KernelProfiler()
    lib.profiler = prof
    nprof = 2
    for _ in range(nprof):  # local steps (no gradient exchange): only this rank profiles, so it must not enter a collective
      trainer.step_count += 1
      trainer.eng.buckets.begin_issue()  # (serial number for the completion signals this pass raises)
      trainer._step_body(batch)
      trainer._optimizer(trainer.step_count)
    lib.profiler = None
    agg = prof.summary()
    total_ms = sum(a['ms'] for a in agg.values())
    peak = PEAK_TFLOPS[args.dtype]
    ridge = peak * 1e12 / (PEAK_HBM_GBS * 1e9)  # FLOP per byte above which the MFMA peak, not HBM, bounds a kernel

    def roof_of(fam, a):
      """roofline object of one kernel family (bench.py docstring): bound decided by the family's algorithmic FLOP/byte against the ridge."""
      mfma = a['bytes'] <= 0 or a['flops'] / a['bytes'] >= ridge
      traffic, traffic_note = pmc_traffic(fam)
      if mfma:
        ach, pk, unit = a['flops'] / (a['ms'] * 1e-3) / 1e12, peak, 'TFLOP/s'
      else:
        ach, pk, unit = a['bytes'] / (a['ms'] * 1e-3) / 1e9, PEAK_HBM_GBS, 'GB/s'
      return {'bound': 'mfma' if mfma else 'hbm', 'kernel': fam, 'achieved': round(ach, 2), 'peak': pk, 'unit': unit, 'frac': round(ach / pk, 4),
              'graph_trace': graph_trace_of(gtrace, fam, a['flops'] / a['calls'], (a['bytes'] / a['calls']) if a.get('bytes') else 0.0, mfma, pk),
              'mfma_counters': mfma_counters_of(fam),
              'traffic': traffic, 'traffic_note': traffic_note,
              'traffic_over_algorithmic_bytes': round(traffic / (a['bytes'] / a['calls']), 2) if (traffic and a.get('bytes')) else None,
              'algorithmic_flop_per_launch': round(a['flops'] / a['calls']),
              'algorithmic_bytes_per_launch': round(a['bytes'] / a['calls']) if a.get('bytes') else None,
              'algorithmic_flop_per_byte': round(a['flops'] / a['bytes'], 1) if a.get('bytes') else None,
              'launches_per_step': a['calls'] // nprof, 'avg_launch_us': round(1e3 * a['ms'] / a['calls'], 2),
              'share_of_step_kernel_time': round(a['ms'] / total_ms, 3)}

    fam, a = max(((f, a) for f, a in agg.items() if a['flops'] > 0), key=lambda fa: fa[1]['ms'])
    roof = roof_of(fam, a)
    # the dominant MFMA-bound family as well (north_star prices the fusion-transformer GEMMs against the MFMA peak); the same object
    # when the dominant family is itself MFMA-bound
    mf = [(f, x) for f, x in agg.items() if x['flops'] > 0 and (x['bytes'] <= 0 or x['flops'] / x['bytes'] >= ridge)]
    roof_mfma = roof_of(*max(mf, key=lambda fa: fa[1]['ms'])) if mf else None
    # ... and the dominant HBM-bound GEMM family (the two lead families are within 1 % of each other in time, so which of them is "the"
    # dominant kernel flips run to run: both are always on the line), plus the north-star kernels by name (fusion-transformer linears)
    hb = [(f, x) for f, x in agg.items() if x['flops'] > 0 and x['bytes'] > 0 and x['flops'] / x['bytes'] < ridge]
    roof_hbm = roof_of(*max(hb, key=lambda fa: fa[1]['ms'])) if hb else None
    # the north-star GEMMs BY LAYER, whatever kernel runs them: every forward / data-gradient linear of the four fusion transformers
    # (transfuser.py:352-359,383-402; KernelProfiler group 'fusion_linears')
    groups = prof.summary(by_group=True)
    grp = groups.get('fusion_linears_c1512')  # the stage-4 transformer (n_embd 1512): 96 % of the fusion-linear FLOPs
    if grp is not None:
      roof_fusion = roof_of('fusion-transformer linears, n_embd = 1512 (forward + data gradient)', grp)
      roof_fusion['kernels'] = {k: v // nprof for k, v in sorted(grp['kernels'].items())}
      roof_fusion['traffic'], roof_fusion['traffic_note'] = pmc_traffic('conv_gemm<bf16,glds256x128>')
      roof_fusion['mfma_counters'] = mfma_counters_of('conv_gemm<bf16,glds256x128>')
      allg = groups.get('fusion_linears')
      if allg is not None:  # ... and all four scales (n_embd 72 / 216 / 576 / 1512) together
        roof_fusion['all_scales'] = {'achieved': round(allg['flops'] / (allg['ms'] * 1e-3) / 1e12, 2), 'frac': round(allg['flops'] / (allg['ms'] * 1e-3) / 1e12 / peak, 4),
                                     'launches_per_step': allg['calls'] // nprof, 'kernels': {k: v // nprof for k, v in sorted(allg['kernels'].items())}}
      roof_fusion['isolated'] = ('profiles/r04_gemm_pp_micro.txt: the same shapes alone on the chip, ring kernels and the round-4 ping-pong GEMM (0.39-0.42 isolated, slower in the '
                                 'step in rounds 4 and 5 -- profiles/r05_ab_gemm_pp_in_step.txt -- and removed in round 5)')
    # the whole step against both roofs: sum of algorithmic FLOPs (every GEMM / attention launch) and of the GEMM families' algorithmic bytes over
    # the MEASURED step time of the timed region above (hipGraph replay), launch count, kernel time
    flop_step = sum(x['flops'] for x in agg.values()) / nprof
    bytes_step = sum(x['bytes'] for x in agg.values()) / nprof
    step_s = elapsed / args.steps
    roof_step = {'algorithmic_tflop': round(flop_step / 1e12, 4), 'algorithmic_gb_gemm_families': round(bytes_step / 1e9, 3),
                 'ms_per_step': round(1e3 * step_s, 3), 'achieved_tflops': round(flop_step / step_s / 1e12, 2), 'frac_of_mfma_peak': round(flop_step / step_s / 1e12 / peak, 4),
                 'at_mfma_peak_ms': round(1e3 * flop_step / (peak * 1e12), 3), 'at_hbm_peak_ms_gemm_families': round(1e3 * bytes_step / (PEAK_HBM_GBS * 1e9), 3),
                 'launches_per_step_library_calls': sum(x['calls'] for x in agg.values()) // nprof,
                 'sum_kernel_ms_eager_events': round(total_ms / nprof, 3),
                 'note': 'FLOPs: every GEMM-shaped and attention launch (2MNK / 4BhT^2d ...); bytes: operands of the GEMM families read once + result written once '
                         '(elementwise / normalisation passes carry no tag: their traffic is the gap the fusions of DESIGN.md close)'}
    if gtrace is not None:
      qs = sorted(gtrace.get('queues', {}).values(), key=lambda q: -q['busy_ms'])
      roof_step['graph_trace'] = {'source': gtrace['source'], 'launches': gtrace['launches'], 'span_ms': gtrace['span_ms'], 'sum_kernel_ms': gtrace['sum_kernel_ms'],
                                  'busy_union_ms': gtrace['busy_union_ms'], 'critical_queue_busy_ms': qs[0]['busy_ms'] if qs else None,
                                  'queues_busy_ms': [q['busy_ms'] for q in qs]}
    if args.kernel_table:
      for f, x in sorted(agg.items(), key=lambda fa: -fa[1]['ms']):
        tf = x['flops'] / (x['ms'] * 1e-3) / 1e12 if x['flops'] else 0.0
        print(f'{f:42s} calls/step {x["calls"] // nprof:5d}  ms/step {x["ms"] / nprof:9.3f}  {100 * x["ms"] / total_ms:5.1f}%  {tf:8.1f} TFLOP/s',
              file=sys.stderr)
  dropin = None
  if not args.no_dropin:
    # the same kernels behind the boundary BASELINE.json::north_star names (LidarCenterNet.forward + train.py's loop), every rank takes part
    try:
      dropin = {'fused_optimizer': dropin_step_time(model, cfg, batch, args.steps, args.warmup, 'fused', log, ddp=rccl_ranks is not None)}
      if rccl_ranks is None:
        dropin['torch_adamw'] = dropin_step_time(model, cfg, batch, max(3, args.steps // 2), 4, 'torch', log)
      dropin['samples_per_s'] = round(args.batch_size * world / (dropin['fused_optimizer']['ms_per_step'] * 1e-3), 1)
      dropin['note'] = ('module driven as team_code/train.py:776-910 drives the reference (DistributedDataParallel when ranks > 1, compute_loss, '
                        '.item() per loss, loss.backward(), optimizer.step(), zero_grad(set_to_none=True)); fused_optimizer = carla_garage_amd.optim.FlatAdamW '
                        'in place of optim.AdamW (INTEGRATION.md), torch_adamw = the unmodified optimizer.  Covered by GPU tests on this boundary: ZeroRedundancyOptimizer, '
                        'freeze_backbone, validate(), grad clip, learnable loss weights, GradScaler (tests/test_boundary_gpu.py); SyncBatchNorm conversion (config.sync_batch_norm = 1; '
                        'the reference default is 0) runs eagerly with the statistics all-reduced over the ranks (tests/test_dist_gpu.py, two ranks on one GPU)')
    except Exception as e:  # pylint: disable=broad-except
      log(f'drop-in leg failed: {type(e).__name__}: {e}')
      dropin = {'error': f'{type(e).__name__}: {e}'}
  fp32_leg = None
  if rank == 0 and rccl_ranks is None and args.dtype == 'bf16' and not args.no_inference:
    # the reference's own arithmetic (fp32; TF32 on the authors' GPUs) on the same step, for context beside the bf16 headline (no TF32 on gfx950:
    # exact-fp32 MFMA, 157 TFLOP/s peak)
    try:
      from carla_garage_amd.graph import GraphedTrainStep
      cfg32 = GlobalConfig(tfpp_dtype='fp32')
      torch.manual_seed(0)
      m32 = LidarCenterNet(cfg32).to(device).train()
      tr32 = Trainer(m32, lr=cfg32.lr)
      g32 = GraphedTrainStep(tr32, batch, warmup=1)
      for _ in range(2):
        g32()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      n32 = max(3, args.steps // 3)
      for _ in range(n32):
        v32 = g32()
      torch.cuda.synchronize()
      ms32 = 1e3 * (time.perf_counter() - t1) / n32
      fp32_leg = {'ms_per_step': round(ms32, 3), 'samples_per_s': round(args.batch_size / (ms32 * 1e-3), 1), 'dtype': 'fp32', 'steps': n32,
                  'final_weighted_loss': round(float(tr32.total_loss(v32)), 5)}
      log(f'fp32 step bs={args.batch_size}: {fp32_leg}')
      final_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
      fp32_leg['bf16_vs_fp32_gradients'] = bf16_vs_fp32_gradients(batch, device, log, final_sd)
      fp32_leg['bf16_vs_fp32_gradients_at_init'] = bf16_vs_fp32_gradients(batch, device, log)
      if not args.no_bf16_reference:
        # is the distance between the bf16 step and the fp32 step inherent to bf16 on THESE weights, or a kernel's fault?  The
        # same statistics for a bf16 reference (the oracle under CPU autocast) on the same weights and batch; at initialisation too
        bf16_ref = {'hip_bf16_vs_hip_fp32': fp32_leg['bf16_vs_fp32_gradients'], 'cpu_autocast_vs_cpu_fp32': autocast_reference(final_sd, args.batch_size, 1234 + rank, log),
                    'what': 'gradients of one train step (bs = 12, same weights = as left by the timed run, same batch, same dropout masks within each pair): the HIP '
                            'bf16 step against the HIP fp32 step, and the CPU oracle under torch.autocast(bfloat16) -- what train.py:885 would run -- against the '
                            'oracle in fp32; formulas of oracle/grad_stats.py.  The HIP step is pinned to be no worse than the unmodified reference under autocast on '
                            'the deterministic test weights (tests/golden/tfpp_bf16_autocast_bs12.npz, tests/test_model.py)'}
        try:
          import numpy as np
          gfix = np.load(os.path.join(ROOT, 'tests', 'golden', 'tfpp_bf16_autocast_bs12.npz'))
          bf16_ref['reference_autocast_vs_reference_fp32_test_weights'] = {str(k): round(float(v), 5) for k, v in zip(gfix['stat_names'], gfix['stats_autocast_vs_fp32'])}
        except Exception:  # pylint: disable=broad-except
          pass
        fp32_leg['bf16_vs_autocast_reference'] = bf16_ref
        log(f'bf16 vs autocast reference: {bf16_ref}')
      del final_sd
      del g32, tr32, m32
      torch.cuda.empty_cache()
    except Exception as e:  # pylint: disable=broad-except
      log(f'fp32 leg failed: {type(e).__name__}: {e}')
  fwd = lidar_hist = swin_fwd = image_aug = None
  if rank == 0 and not args.no_inference:
    fwd = inference_latency(model, cfg, device, log)
    fwd.update(forward_call_latency(device, log))  # the module's own capture (the default), measured in an agent-like child process
    lidar_hist = lidar_histogram_latency(cfg, device, log)
    try:
      image_aug = image_augmentation_latency(cfg, device, log, bs=args.batch_size)
    except Exception as e:  # pylint: disable=broad-except
      log(f'image augmentation leg failed: {type(e).__name__}: {e}')
    swin_fwd = video_swin_forward(device, log, train=rccl_ranks is None)
  if rccl_ranks is not None:
    _flush_c_stdio()  # (RCCL prints a version banner through C stdio: buffered on a pipe, it would otherwise appear at exit, AFTER the JSON line)
    dist.barrier()

  line = None
  if rank == 0:
    ms = 1e3 * elapsed / args.steps
    value = args.batch_size * world * args.steps / elapsed
    line = {
        'metric': 'training samples/sec (TransFuser++ bs=12/GPU)', 'value': round(value, 3), 'unit': 'samples/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'TransFuser++ training step (fwd + 10 losses + bwd + AdamW-amsgrad), RegNetY-3.2GF x2, 256x1024 RGB + '
                               '256x256 LiDAR BEV, train-mode BN + dropout, random-init weights',
                   'global_batch': args.batch_size * world, 'per_gpu_batch': args.batch_size, 'parallelism': f'dp{world}',
                   'algorithmic_tflop_per_step_per_gpu': round(GFLOP_PER_SAMPLE_TRAIN * args.batch_size / 1e3, 3),
                   'model_tflops_per_gpu': round(GFLOP_PER_SAMPLE_TRAIN * args.batch_size / 1e3 / (ms * 1e-3), 2),
                   'final_weighted_loss': round(loss_total, 5), 'hipgraph_step': graphed},
    }
    if fwd is not None:
      line['fwd_ms_per_frame'] = fwd
    if lidar_hist is not None:
      line['lidar_histogram_60k_points_us'] = lidar_hist
    if image_aug is not None:
      line['image_augmentation'] = image_aug
    if swin_fwd is not None:
      line['video_swin_forward_bs4'] = swin_fwd
    if comm is not None:
      line['gradient_exchange'] = comm
    if dropin is not None:
      line['dropin'] = dropin
    if fp32_leg is not None:
      if 'bf16_vs_autocast_reference' in fp32_leg:
        line['bf16_vs_autocast_reference'] = fp32_leg.pop('bf16_vs_autocast_reference')
      line['fp32_step'] = fp32_leg
    if roof_step is not None:
      line['roofline_step'] = roof_step
    if roof is not None:
      line['roofline'] = roof
      if roof_mfma is not None and roof_mfma['kernel'] != roof['kernel']:
        line['roofline_mfma'] = roof_mfma
      if roof_hbm is not None and roof_hbm['kernel'] != roof['kernel']:
        line['roofline_hbm'] = roof_hbm
      if roof_fusion is not None and roof_fusion['kernel'] != roof['kernel']:
        line['roofline_fusion_linears'] = roof_fusion
    line['parity'] = {'fp32': 'outputs / losses 1e-3 (measured 3e-6), per-parameter gradient norms 1e-2, against goldens written by the unmodified reference '
                              '(tests/golden, oracle/make_golden.py)',
                      'bf16': 'pinned against a bf16 reference: the HIP bf16 step (vs the HIP fp32 step) is no worse on any gradient statistic than the unmodified '
                              'reference under torch.autocast(bfloat16) (vs its fp32 step) -- tests/golden/tfpp_bf16_autocast_bs12.npz, oracle/make_golden_bf16.py; '
                              'bf16_vs_autocast_reference on this line repeats the comparison on the weights this run ended with',
                      'unpinned_third_party': ['timm 0.6.7 (absent: RegNetY block arithmetic pinned bit-exactly against HF transformers instead)',
                                               'shapely (absent: rotated IoU pinned against an exact rational-arithmetic oracle)',
                                               'imgaug 0.4.0 / opencv 4.6 (absent: operator arithmetic restated in oracle/imgaug_port.py)']}
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline()
  torch.cuda.synchronize()  # nothing in flight when the graphs and the arenas are torn down
  if rccl_ranks is not None:
    dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0 and world > 1:
      time.sleep(0.5)  # (the other ranks' last buffered output reaches the launcher's pipe first)
  if line is not None:
    sys.stdout.flush()
    print(json.dumps(line), flush=True)  # the ONE JSON line is the last thing this job writes to stdout


if __name__ == '__main__':
  main()
