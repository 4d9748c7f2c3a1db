"""Host -> device batch path of the training loop (team_code/train.py:688-766, ``Engine.load_data_compute_loss``): the reference moves every
tensor of a collated ``CARLA_Data`` batch with a synchronous ``.to(device, dtype=...)`` from pageable memory, ~70 MB per step at bs = 12.

``DeviceBatchPrefetcher`` wraps any iterable of such host batches (the reference's own DataLoader, or tools/reference_train_shim.py's synthetic
dataset) and yields device batches in the layout ``Trainer.train_step`` takes: the tensors are staged in pinned buffers (uint8 images / semantic
maps and int32 targets keep their host dtype over PCIe and are widened on the GPU by tfpp_widen) and uploaded on a dedicated HIP
stream into one of two device slots, so the upload of batch i+1 overlaps the compute of step i; an event hands the slot to the compute
stream, and a slot is only overwritten after the step that consumed it has been enqueued (event on the compute stream)."""
import queue
import threading

import torch

from . import losses, ops
from ._lib import lib
from .ops import ptr

# reference batch key -> (Trainer batch key, device dtype, needed when)
_FLOAT, _LONG = torch.float32, torch.int64
KEYMAP = (
    ('rgb', 'rgb', _FLOAT, lambda c: True),
    ('lidar', 'lidar_bev', _FLOAT, lambda c: c.lidar_seq_len == 1),
    ('temporal_lidar', 'lidar_bev', _FLOAT, lambda c: c.lidar_seq_len > 1),
    ('target_point', 'target_point', _FLOAT, lambda c: True),
    ('command', 'command', _FLOAT, lambda c: True),
    ('speed', 'ego_vel', _FLOAT, lambda c: True),
    ('target_speed', 'target_speed_label', _LONG, lambda c: True),
    ('route', 'checkpoint_label', _FLOAT, lambda c: True),
    ('ego_waypoints', 'waypoint_label', _FLOAT, lambda c: c.use_wp_gru),
    ('semantic', 'semantic_label', _LONG, lambda c: c.use_semantic),
    ('bev_semantic', 'bev_semantic_label', _LONG, lambda c: c.use_bev_semantic),
    ('depth', 'depth_label', _FLOAT, lambda c: c.use_depth),
    ('center_heatmap', 'center_heatmap_label', _FLOAT, lambda c: c.detect_boxes),
    ('wh', 'wh_label', _FLOAT, lambda c: c.detect_boxes),
    ('yaw_class', 'yaw_class_label', _LONG, lambda c: c.detect_boxes),
    ('yaw_res', 'yaw_res_label', _FLOAT, lambda c: c.detect_boxes),
    ('offset', 'offset_label', _FLOAT, lambda c: c.detect_boxes),
    ('velocity', 'velocity_label', _FLOAT, lambda c: c.detect_boxes and losses.temporal(c)),
    ('brake_target', 'brake_target_label', _LONG, lambda c: c.detect_boxes and losses.temporal(c)),
    ('pixel_weight', 'pixel_weight_label', _FLOAT, lambda c: c.detect_boxes),
    ('avg_factor', 'avg_factor_label', _FLOAT, lambda c: c.detect_boxes),
)


_NARROW = {torch.uint8: 0, torch.int32: 1}  # host dtypes that travel as they are and are widened by tfpp_widen
_WIDE = {torch.float32: 0, torch.int64: 1}


class DeviceBatchPrefetcher:
  """for batch in DeviceBatchPrefetcher(loader, config): trainer.train_step(batch)   (or a GraphedTrainStep)

  A staging thread pulls host batches from ``loader``, converts them into the pinned buffers of a free slot and enqueues the upload (+ the
  widening kernels) on the copy stream; the consumer thread only waits for the slot's event on its compute stream.  A yielded batch stays
  valid until the consumer asks for the next one AND the work it enqueued meanwhile on the current stream has run."""

  def __init__(self, loader, config, device='cuda', slots=2, rasterise_on_device=False, augment=None, lidar_on_device=False):
    """augment: a carla_garage_amd.augment.ImageAugmenter -- the colour augmentation of team_code/data.py:481-496,1141-1157 runs on the
    uploaded uint8 frame (copy stream) instead of in the loader's workers; the loader then yields the frame as decoded (use_color_aug = 0).
    rasterise_on_device: the host batches carry ``bounding_boxes_f64`` (B, n, 8) float64 + ``num_bounding_boxes`` (B,) (collate_boxes)
    instead of the nine CenterNet label maps, which are then drawn on the GPU by rasterise_targets right after the upload.
    lidar_on_device (round 5): the host batches carry the RAW sweeps -- ``lidar_sweeps``: a list of B x T float64 (N_i, 3) arrays in
    (sample, time frame) order, as laspy's .xyz yields them (data.py:365), and ``lidar_align``: (B x T, 10) float64 from
    carla_garage_amd.lidar.align_params (collate_lidar below builds both) -- instead of the BEV images ``lidar`` / ``temporal_lidar``;
    CARLA_Data.align + lidar_to_histogram_features (data.py:524-560,840-906: ~4 ms of numpy per frame in the loader's workers) then run on
    the copy stream (tfpp_lidar_align_histogram), bit-exactly."""
    self.loader, self.cfg, self.device = loader, config, torch.device(device)
    self.rasterise = bool(rasterise_on_device and config.detect_boxes)
    self.augment = augment
    self.lidar_on_device = bool(lidar_on_device)
    self._hist = None
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    from . import streams
    self.copy_stream = streams.get(self.device, 'copy')
    self.slots = [dict(pin={}, dev={}, ready=torch.cuda.Event(), used=False) for _ in range(slots)]
    self.keys = [(src, dst, dt) for src, dst, dt, need in KEYMAP if need(config) and not (self.rasterise and dst in TARGET_KEYS) and
                 not (self.lidar_on_device and dst == 'lidar_bev')]
    if self.rasterise:
      self.keys += [('bounding_boxes_f64', 'bounding_boxes_f64', torch.float64), ('num_bounding_boxes', 'num_bounding_boxes', torch.int32)]

  def _host_view(self, src, t):
    if not torch.is_tensor(t):
      t = torch.as_tensor(t)
    if src == 'route':  # train.py:747: data['route'][:, :predict_checkpoint_len]
      t = t[:, :self.cfg.predict_checkpoint_len]
    if src == 'speed':  # train.py:726: .unsqueeze(1)
      t = t.reshape(-1, 1)
    return t

  def _upload(self, slot, released, host_batch):
    """Stage one host batch into ``slot`` and enqueue its upload on the copy stream (staging thread)."""
    if slot['used']:
      slot['ready'].synchronize()  # host side: the previous upload out of this slot's pinned buffers has finished
    if released is not None:
      self.copy_stream.wait_event(released)  # device side: the step that read this slot was enqueued before that event
    out = {}
    with torch.cuda.stream(self.copy_stream):
      for src, dst, dt in self.keys:
        t = self._host_view(src, host_batch[src])
        narrow = t.dtype in _NARROW and t.dtype != dt  # 1 or 4 bytes per value over PCIe, widened on the device
        stage_dt = t.dtype if narrow else dt
        pin = slot['pin'].get(src)
        if pin is None or pin.shape != t.shape or pin.dtype != stage_dt:
          pin = slot['pin'][src] = torch.empty(t.shape, dtype=stage_dt, pin_memory=True)
          slot['dev'][src] = torch.empty(t.shape, dtype=stage_dt, device=self.device)
          slot['dev'][src + '/wide'] = torch.empty(t.shape, dtype=dt, device=self.device) if narrow else None
        pin.copy_(t)  # host-side conversion / gather into pinned memory
        d = slot['dev'][src]
        d.copy_(pin, non_blocking=True)
        if self.augment is not None and src == 'rgb':
          if d.dtype != torch.uint8:
            raise ValueError('device-side augmentation works on the decoded uint8 frame: the loader must yield rgb as uint8')
          d = self.augment.apply(d)
        if narrow:
          wide = slot['dev'][src + '/wide']
          lib.tfpp_widen(ptr(d), ptr(wide), d.numel(), _NARROW[t.dtype], _WIDE[dt], ops.stream())
          d = wide
        out[dst] = d
      if self.lidar_on_device:
        out['lidar_bev'] = self._lidar_bev(slot, host_batch)
      if self.rasterise:
        boxes, counts = out.pop('bounding_boxes_f64'), out.pop('num_bounding_boxes')
        tg = slot['dev'].get('/targets')
        if tg is not None and tg['wh_label'].shape[0] != boxes.shape[0]:
          tg = None
        slot['dev']['/targets'] = tg = rasterise_targets(boxes, counts, self.cfg, out=tg)
        out.update(tg)
      slot['ready'].record(self.copy_stream)
    slot['used'] = True
    return out

  def _lidar_bev(self, slot, host_batch):
    """Raw sweeps of the batch -> pinned staging -> device -> aligned BEV histograms (B, T * C, H, W), all on the copy stream."""
    from .lidar import LidarBatchHistogram
    import numpy as np
    if self._hist is None:
      self._hist = LidarBatchHistogram(self.cfg, self.device)
    sweeps, xf = host_batch['lidar_sweeps'], torch.as_tensor(host_batch['lidar_align'], dtype=torch.float64).reshape(-1, 10)
    frames = len(sweeps)
    T = max(1, int(self.cfg.lidar_seq_len))
    if frames % T or xf.shape[0] != frames:
      raise ValueError(f'lidar_sweeps: {frames} sweeps / {xf.shape[0]} parameter rows for lidar_seq_len = {T}')
    sizes = [int(np.asarray(s).reshape(-1, 3).shape[0]) for s in sweeps]
    total = sum(sizes)
    cap = slot['pin'].get('/lidar_cap', 0)
    if total > cap or slot['pin'].get('/lidar_frames') != frames:
      cap = max(total + total // 4, 1024)
      slot['pin'].update({'/lidar_cap': cap, '/lidar_frames': frames, '/lidar_pts': torch.empty((cap, 3), dtype=torch.float64, pin_memory=True),
                          '/lidar_off': torch.empty(frames + 1, dtype=torch.int64, pin_memory=True), '/lidar_xf': torch.empty((frames, 10), dtype=torch.float64, pin_memory=True)})
      slot['dev'].update({'/lidar_pts': torch.empty((cap, 3), dtype=torch.float64, device=self.device), '/lidar_off': torch.empty(frames + 1, dtype=torch.int64, device=self.device),
                          '/lidar_xf': torch.empty((frames, 10), dtype=torch.float64, device=self.device), '/lidar_bev': None})
    pin, dev = slot['pin'], slot['dev']
    pts_np, off_np = pin['/lidar_pts'].numpy(), pin['/lidar_off'].numpy()
    o = 0
    for i, (s_, n) in enumerate(zip(sweeps, sizes)):
      off_np[i] = o
      if n:
        pts_np[o:o + n] = np.asarray(s_, dtype=np.float64).reshape(-1, 3)
      o += n
    off_np[frames] = o
    pin['/lidar_xf'].copy_(xf)
    dev['/lidar_pts'][:max(total, 1)].copy_(pin['/lidar_pts'][:max(total, 1)], non_blocking=True)
    dev['/lidar_off'].copy_(pin['/lidar_off'], non_blocking=True)
    dev['/lidar_xf'].copy_(pin['/lidar_xf'], non_blocking=True)
    gp = bool(self.cfg.use_ground_plane)
    dev['/lidar_bev'] = self._hist.from_device(dev['/lidar_pts'], dev['/lidar_off'], dev['/lidar_xf'], frames, gp, out=dev['/lidar_bev'], total_points=total)
    bev = dev['/lidar_bev']
    return bev.view(frames // T, T * bev.shape[1], bev.shape[2], bev.shape[3])  # (b, t) frame order = the channel concatenation of data.py:536,558

  def _stager(self, free_q, ready_q, stop):
    try:
      torch.cuda.set_device(self.device)
      torch.set_num_threads(1)  # per-thread OpenMP setting: a second 256-thread team for 30 MB of memcpy costs more than it saves
      for host_batch in self.loader:
        item = free_q.get()
        if item is None or stop.is_set():
          return
        i, released = item
        ready_q.put((i, self._upload(self.slots[i], released, host_batch)))
      ready_q.put(None)
    except BaseException as e:  # pylint: disable=broad-except
      ready_q.put(e)

  def __iter__(self):
    free_q, ready_q, stop = queue.Queue(), queue.Queue(), threading.Event()
    for i in range(len(self.slots)):
      free_q.put((i, None))
    th = threading.Thread(target=self._stager, args=(free_q, ready_q, stop), daemon=True)
    th.start()
    try:
      while True:
        item = ready_q.get()
        if item is None:
          break
        if isinstance(item, BaseException):
          raise item
        i, batch = item
        torch.cuda.current_stream(self.device).wait_event(self.slots[i]['ready'])
        yield batch
        # the consumer has enqueued its step on the current stream by the time it asks for the next batch
        released = torch.cuda.Event()
        released.record(torch.cuda.current_stream(self.device))
        free_q.put((i, released))
    finally:
      stop.set()
      free_q.put(None)
      # the stager may be blocked inside `for host_batch in self.loader` (a slow or stalled DataLoader) where it cannot see `stop`: it is a
      # daemon thread, so an early exit of the consumer (break / exception) must not wait for the next host batch indefinitely
      th.join(timeout=5.0)


TARGET_KEYS = ('center_heatmap_label', 'wh_label', 'offset_label', 'yaw_class_label', 'yaw_res_label', 'velocity_label', 'brake_target_label',
               'pixel_weight_label', 'avg_factor_label')
MIN_OVERLAP = 0.1  # data.py:753


def rasterise_targets(boxes, counts, config, out=None):
  """CenterNet label maps of a batch on the GPU (tfpp_centernet_targets; replaces the per-sample CARLA_Data.get_targets of the loader
  workers, team_code/data.py:588-600,697-790).  boxes: (B, max_boxes, 8) float64 device tensor of parse_bounding_boxes rows, counts: (B,)
  int32 valid rows per sample.  Returns the nine ``*_label`` tensors in the trainer's layout (``out``: a dict of them to overwrite)."""
  if not (boxes.is_cuda and boxes.dtype == torch.float64 and boxes.dim() == 3 and boxes.shape[2] == 8 and boxes.is_contiguous()):
    raise ValueError('rasterise_targets: boxes must be a contiguous (B, max_boxes, 8) float64 device tensor')
  if not (counts.is_cuda and counts.dtype == torch.int32 and counts.shape == (boxes.shape[0],)):
    raise ValueError('rasterise_targets: counts must be a (B,) int32 device tensor')
  B, nb = boxes.shape[0], boxes.shape[1]
  H = config.lidar_resolution_height // config.bev_down_sample_factor
  W = config.lidar_resolution_width // config.bev_down_sample_factor
  C = config.num_bb_classes
  if out is None:
    f = dict(device=boxes.device, dtype=torch.float32)
    i = dict(device=boxes.device, dtype=torch.int64)
    out = dict(center_heatmap_label=torch.empty((B, C, H, W), **f), wh_label=torch.empty((B, 2, H, W), **f),
               offset_label=torch.empty((B, 2, H, W), **f), yaw_class_label=torch.empty((B, H, W), **i),
               yaw_res_label=torch.empty((B, 1, H, W), **f), velocity_label=torch.empty((B, 1, H, W), **f),
               brake_target_label=torch.empty((B, H, W), **i), pixel_weight_label=torch.empty((B, 2, H, W), **f),
               avg_factor_label=torch.empty((B,), **f))
  o = out
  lib.tfpp_centernet_targets(ptr(boxes), ptr(counts), ptr(o['center_heatmap_label']), ptr(o['wh_label']), ptr(o['offset_label']),
                             ptr(o['yaw_class_label']), ptr(o['yaw_res_label']), ptr(o['velocity_label']), ptr(o['brake_target_label']),
                             ptr(o['pixel_weight_label']), ptr(o['avg_factor_label']), B, nb, C, H, W, config.num_dir_bins,
                             float(W / config.lidar_resolution_width), float(H / config.lidar_resolution_height), MIN_OVERLAP, ops.stream())
  return out


def collate_boxes(box_lists, max_boxes=None):
  """Host-side collation for rasterise_on_device: a list (one entry per sample) of the float64 (n_i, 8) arrays parse_bounding_boxes returns
  (team_code/data.py:565-570, BEFORE the float32 padding of data.py:571-585) -> (boxes (B, max_i n_i, 8) float64, counts (B,) int32)."""
  import numpy as np
  arrs = [np.asarray(b, dtype=np.float64).reshape(-1, 8) for b in box_lists]
  n = max([a.shape[0] for a in arrs] + [1]) if max_boxes is None else max_boxes
  boxes = np.zeros((len(arrs), n, 8), np.float64)
  counts = np.zeros(len(arrs), np.int32)
  for i, a in enumerate(arrs):
    if a.shape[0] > n:
      raise ValueError(f'collate_boxes: sample {i} has {a.shape[0]} boxes, max_boxes = {n}')
    boxes[i, :a.shape[0]] = a
    counts[i] = a.shape[0]
  return torch.from_numpy(boxes), torch.from_numpy(counts)


def to_reference_batch(batch, config, rgb_uint8=True):
  """Inverse key mapping: a Trainer-layout host batch (bench.synthetic_batch(device=None)) as the reference's collated CARLA_Data
  dict (data.py:__getitem__ names and host dtypes), for tests and tools that feed DeviceBatchPrefetcher."""
  out = {}
  for src, dst, dt, need in KEYMAP:
    if not need(config) or dst not in batch:
      continue
    t = batch[dst]
    if rgb_uint8 and src in ('rgb', 'semantic', 'bev_semantic'):  # decoded images: data.py:511-522
      t = t.to(torch.uint8)
    elif rgb_uint8 and src in ('yaw_class', 'brake_target'):  # data.py:725-728
      t = t.to(torch.int32)
    if src == 'speed':
      t = t.reshape(-1)
    out[src] = t
  return out


def collate_lidar(samples, config):
  """Host-side collation for lidar_on_device: ``samples`` = one dict per dataset item carrying ``lidar_sweeps`` (the T = lidar_seq_len raw float64
  sweeps of the item, oldest first, as laspy's .xyz returns them -- team_code/data.py:365) and ``lidar_align`` ((T, 10) float64, one
  carla_garage_amd.lidar.align_params(measurements_i, target, y_augmentation, yaw_augmentation) row per sweep with the target frame of
  data.py:524-553) next to the other CARLA_Data keys.  Returns the batch dict DeviceBatchPrefetcher(lidar_on_device=True) takes: every other
  key stacked as torch's default collate stacks it, the sweeps as a flat list in (sample, time) order, the parameters as (B x T, 10)."""
  import numpy as np
  from torch.utils.data import default_collate
  rest = [{k: v for k, v in s.items() if k not in ('lidar_sweeps', 'lidar_align', 'lidar', 'temporal_lidar')} for s in samples]
  batch = default_collate(rest) if rest and rest[0] else {}
  batch['lidar_sweeps'] = [np.asarray(sw, dtype=np.float64).reshape(-1, 3) for s in samples for sw in s['lidar_sweeps']]
  batch['lidar_align'] = torch.from_numpy(np.concatenate([np.asarray(s['lidar_align'], dtype=np.float64).reshape(-1, 10) for s in samples], axis=0))
  T = max(1, int(config.lidar_seq_len))
  if len(batch['lidar_sweeps']) != len(samples) * T:
    raise ValueError(f'collate_lidar: expected {T} sweeps per sample (lidar_seq_len)')
  return batch
