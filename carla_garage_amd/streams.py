"""One fixed set of HIP streams per device for the whole process (lanes of the network, weight-gradient lane, weight repack, collectives,
hipGraph capture, loader copies).

Rounds 1-5 created ``torch.cuda.Stream()`` objects per Engine / SideLane / GradBuckets.  PyTorch hands those out round-robin from a pool of 32
native streams per device and priority: a long process (a test session, an agent that builds several models) wraps around the pool, and a
"new" lane stream is then the SAME native stream as one that is already in use -- for instance the capture stream of carla_garage_amd/graph.py.
A capture whose branch stream aliases the capturing stream records a different graph than the code describes, the per-stream scratch tables
of ops.py are shared between lanes that believe they are concurrent, and the first replay of such a capture crashed inside hipGraphLaunch
(round 5: "history-dependent" segfault of the module's own eval capture; round 6: reproduced with three tests in a row, gone with
TFPP_BRANCH_STREAMS=0, gone with this registry).  Here every role gets its stream once; the registry checks that the native handles are
pairwise distinct.  Engines of several models share the streams: the host issues their passes one after the other, stream order is kept."""
import torch

_STREAMS = {}


def _dev_key(device):
  device = torch.device(device)
  idx = device.index if device.index is not None else torch.cuda.current_device()
  return idx


def get(device, role, index=0):
  """The process-wide stream of (device, role, index); roles: 'lane', 'side', 'pack', 'comm', 'capture', 'copy'."""
  d = _dev_key(device)
  key = (d, role, index)
  st = _STREAMS.get(key)
  if st is None:
    taken = {s.cuda_stream for (dd, _, _), s in _STREAMS.items() if dd == d}
    for _ in range(64):  # (the pool is round-robin: a handle that another role already holds is skipped)
      st = torch.cuda.Stream(torch.device('cuda', d))
      if st.cuda_stream not in taken:
        break
    else:
      raise RuntimeError('carla_garage_amd.streams: could not obtain a native stream that no other role of this device uses')
    _STREAMS[key] = st
  return st


def handles(device):
  """{(role, index): native handle} of the streams handed out on ``device`` (tests)."""
  d = _dev_key(device)
  return {(r, i): s.cuda_stream for (dd, r, i), s in _STREAMS.items() if dd == d}
