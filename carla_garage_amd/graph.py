"""hipGraph capture of the static launch sequences (inference forward; training step body).

Shapes are static and the engine issues only ``libtfpp_hip.so`` launches + hipMemsetAsync on the current stream, so
one forward (~900 launches at bs=1, the 20 Hz closed-loop tick of sensor_agent.py:456-461) or one training step
(~3500 launches) replays as a single hipGraphLaunch.  torch.cuda.CUDAGraph is used for the capture plumbing
(stream + private memory pool); no ATen kernels are captured.
"""
import torch

# Other threads of the process may issue HIP calls while a capture is open (the RCCL watchdog thread of an initialised process
# group polls its events): with the default "global" mode any such call invalidates the capture.  "thread_local" still rejects
# unsafe calls made by the capturing thread itself.
CAPTURE_MODE = 'thread_local'

import contextlib
import gc


@contextlib.contextmanager
def capture(graph, stream, pool=None):
  """``torch.cuda.graph`` with Python's cyclic garbage collector switched off for the duration of the capture.  The engine creates tens of
  thousands of Python objects while a 3000-launch step is captured; a collection triggered in the middle may finalise a CUDAGraph / stream /
  event of an earlier owner (previous model, previous test), whose destructor calls into the HIP runtime from inside the open capture and
  aborts the process (round 4: `Fatal Python error: Aborted ... Garbage-collecting` in a capture of the drop-in step).  Everything
  collectable is collected before the capture starts instead."""
  gc.collect()
  was = gc.isenabled()
  gc.disable()
  try:
    kw = {} if pool is None else {'pool': pool}
    with torch.cuda.graph(graph, stream=stream, capture_error_mode=CAPTURE_MODE, **kw):
      yield
  finally:
    if was:
      gc.enable()


_CAPTURE_STREAMS = {}
import os as _os
_HIGH_PRIORITY_LANES = _os.environ.get('TFPP_LANE_PRIORITY', '0') == '1'  # main lanes on high-priority streams, the weight-gradient lane on a normal one (A/B switch)


def capture_stream(device):
  """One capture stream per device for every graph of this module: the per-stream scratch buffers of ops.py are keyed by stream, so
  all captures share one set, prepared (allocated, statistics rows zeroed) before the capture starts."""
  from . import ops
  device = torch.device(device)
  st = _CAPTURE_STREAMS.get(str(device))
  if st is None:
    st = _CAPTURE_STREAMS[str(device)] = torch.cuda.Stream(device, priority=-1 if _HIGH_PRIORITY_LANES else 0)
  st.wait_stream(torch.cuda.current_stream(device))
  with torch.cuda.stream(st):
    ops.clone_scratch_for_current_stream(device)
  torch.cuda.current_stream(device).wait_stream(st)
  return st


class GraphedForward:
  """``y = GraphedForward(model, rgb, lidar_bev, target_point, ego_vel, command)(...)`` -- eval-mode, no-grad forward."""

  def __init__(self, model, *example_inputs, warmup=2):
    self.model = model
    self.static_in = [x.clone() for x in example_inputs]
    with torch.inference_mode():
      for _ in range(warmup):  # populate weight images / constants outside the capture
        model(*self.static_in)
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    st = capture_stream(self.static_in[0].device)
    with torch.inference_mode(), capture(self.graph, st):
      self.static_out = model(*self.static_in)
    torch.cuda.synchronize()

  def __call__(self, *inputs):
    for dst, src in zip(self.static_in, inputs):
      if dst.data_ptr() != src.data_ptr():
        dst.copy_(src, non_blocking=True)
    self.graph.replay()
    return self.static_out


class GraphedTrainStep:
  """Captures the training step body (repack + forward + losses + backward) for a fixed batch layout; the gradient
  all-reduce and the optimizer launch stay outside the graphs so RCCL is never captured.  With more than one rank the body is
  captured as TWO graphs sharing one memory pool, split where the gradients of the heads and of fusion stage 4 are final
  (Tape.mark): their slice of the arena is all-reduced while the second graph replays."""

  def __init__(self, trainer, batch, warmup=2):
    """NOTE: the ``warmup`` eager steps are real training steps (parameters, optimizer state and ``step_count`` advance)."""
    self.trainer = trainer
    self.static_batch = {k: v.clone() for k, v in batch.items()}
    for _ in range(warmup):
      trainer.train_step(self.static_batch)
    if trainer.step_count == 0:
      raise RuntimeError('GraphedTrainStep: run at least one eager train_step first (warmup >= 1): the capture must not be the call '
                         'that sizes and allocates the scratch buffers')
    torch.cuda.synchronize()
    self.split = trainer.overlap_enabled()
    self.graph = torch.cuda.CUDAGraph()
    self.graph2 = None
    dot = _os.environ.get('TFPP_DEBUG_GRAPH_DOT')  # debugging aid (tools/graph_lists.py): hipGraphDebugDotPrint of the captured step
    if dot:
      self.graph.enable_debug_mode()
    st = capture_stream(trainer.eng.device)
    if self.split:
      with capture(self.graph, st):
        self.vals = trainer._step_part1(self.static_batch)
      self.graph2 = torch.cuda.CUDAGraph()
      with capture(self.graph2, st, pool=self.graph.pool()):
        trainer._step_part2()
    else:
      with capture(self.graph, st):
        self.vals = trainer._step_body(self.static_batch)
    if dot:
      self.graph.debug_dump(dot)
    self.early_opt_in_graph = trainer.early_opt_in_step  # the captured step updates the early slice of the arena itself (trainer._early_optimizer)
    trainer.early_opt_in_step = False  # (the capture executed nothing)
    torch.cuda.synchronize()

  def __call__(self, batch=None):
    tr = self.trainer
    if batch is not None:
      for k, dst in self.static_batch.items():
        src = batch[k]
        if dst.data_ptr() != src.data_ptr():
          dst.copy_(src, non_blocking=True)
    tr.step_count += 1
    if self.early_opt_in_graph:
      tr.upload_hyper(tr.step_count)
    self.graph.replay()
    tr.early_opt_in_step = self.early_opt_in_graph
    early = None
    if self.graph2 is not None:
      early = tr.reduce_early()
      self.graph2.replay()
    tr.finish_step(early)
    return self.vals
