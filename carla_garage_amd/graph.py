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
    cm = torch.cuda.graph(graph, stream=stream, capture_error_mode=CAPTURE_MODE, **kw)
    # capture_begin / capture_end update the CUDA generator's graph-state tensors in place; those are created by the FIRST capture of the
    # process, and if that one ran under torch.inference_mode() (GraphedForward) they are inference tensors that no later capture outside
    # inference mode may touch ("Inplace update to inference tensor outside InferenceMode": round 6, bench.py --config5-only).  Enter and
    # leave the capture with inference mode off; the body keeps the caller's mode.
    with torch.inference_mode(False):
      cm.__enter__()
    try:
      yield
    except BaseException:
      import sys
      with torch.inference_mode(False):
        if not cm.__exit__(*sys.exc_info()):
          raise
    else:
      with torch.inference_mode(False):
        cm.__exit__(None, None, None)
  finally:
    if was:
      gc.enable()


import os as _os


def capture_stream(device):
  """One capture stream per device for every graph of this module: the per-stream scratch buffers of ops.py are keyed by stream, so
  all captures share one set, prepared (allocated, statistics rows zeroed) before the capture starts."""
  from . import ops
  from . import streams
  device = torch.device(device)
  st = streams.get(device, 'capture')
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
  """Captures the training step body (repack + forward + losses + backward) for a fixed batch layout as ONE hipGraph, for one rank and for
  eight alike; the gradient all-reduces and the optimizer launches stay outside the graph so RCCL is never captured.  The captured pass
  raises a completion signal (a device word a kernel node raises to the serial number of the pass: buckets.py, include/tfpp.h tfpp_signal_set) behind the kernels
  that complete each bucket of the gradient arena; after ``graph.replay()`` returns -- the replay is still running --
  Trainer.finish_step() issues the all-reduce of every bucket behind a wait on its signal (tfpp_signal_wait on the collective's stream)."""

  def __init__(self, trainer, batch, warmup=2):
    """NOTE: the eager steps in front of the capture are real training steps (parameters, optimizer state and ``step_count`` advance).
    At least one runs, more until the arenas are in their observed completion order (Trainer.apply_observed_layout: the second pass is the
    first one with the lane's final fork points) and one step has run in that layout."""
    self.trainer = trainer
    if trainer.eng.sync_bn and trainer.world > 1:
      raise RuntimeError('GraphedTrainStep: a SyncBatchNorm step (train.py:511-512) all-reduces BatchNorm statistics inside the pass and cannot be captured; '
                         'use Trainer.train_step')
    self.static_batch = {k: v.clone() for k, v in batch.items()}
    steps = 0
    ready = lambda: trainer.step_count > 0 and trainer.layout_final and trainer.eager_steps_in_layout >= 1
    while steps < warmup or (not ready() and steps < warmup + 4):
      trainer.train_step(self.static_batch)
      steps += 1
      trainer.apply_observed_layout()  # (when this step produced a valid observation; the next eager step then runs in the new layout)
    if trainer.step_count == 0:
      raise RuntimeError('GraphedTrainStep: run at least one eager train_step first: the capture must not be the call that sizes and '
                         'allocates the scratch buffers')
    trainer.layout_final = True  # (frozen from here on: the graph holds pointers into the arenas)
    torch.cuda.synchronize()
    self.graph = torch.cuda.CUDAGraph()
    dot = _os.environ.get('TFPP_DEBUG_GRAPH_DOT')  # debugging aid (tools/graph_lists.py): hipGraphDebugDotPrint of the captured step
    if dot:
      self.graph.enable_debug_mode()
    st = capture_stream(trainer.eng.device)
    with capture(self.graph, st):
      self.vals = trainer._step_body(self.static_batch)
    self.program = trainer.program  # the completion signals of the gradient buckets every replay raises (buckets.py)
    if dot:
      self.graph.debug_dump(dot)
    torch.cuda.synchronize()

  def __call__(self, batch=None):
    tr = self.trainer
    if batch is not None:
      for k, dst in self.static_batch.items():
        src = batch[k]
        if dst.data_ptr() != src.data_ptr():
          dst.copy_(src, non_blocking=True)
    tr.step_count += 1
    tr.eng.buckets.begin_issue()  # the serial number the replay's completion signals will carry (buckets.py): in front of the replay
    self.graph.replay()
    tr.finish_step(self.program)
    return self.vals
