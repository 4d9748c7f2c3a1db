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

  def __init__(self, loader, config, device='cuda', slots=2):
    self.loader, self.cfg, self.device = loader, config, torch.device(device)
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self.copy_stream = torch.cuda.Stream(self.device)
    self.slots = [dict(pin={}, dev={}, ready=torch.cuda.Event(), used=False) for _ in range(slots)]
    self.keys = [(src, dst, dt) for src, dst, dt, need in KEYMAP if need(config)]

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
        if narrow:
          wide = slot['dev'][src + '/wide']
          lib.tfpp_widen(ptr(d), ptr(wide), d.numel(), _NARROW[t.dtype], _WIDE[dt], ops.stream())
          d = wide
        out[dst] = d
      slot['ready'].record(self.copy_stream)
    slot['used'] = True
    return out

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
      th.join()


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
