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
import ctypes
import os

import torch

_STREAMS = {}
# TFPP_SIDE_CU_MASK=<CUs per XCD, 1..32> (experiment): the streams of the weight-gradient lane are created with
# hipExtStreamCreateWithCUMask so that their kernels only run on the first n CUs of every XCD.  Measured in round 6 (profiles/r06_ab_side_cu_mask.txt).
_SIDE_CUS = int(os.environ.get('TFPP_SIDE_CU_MASK', '0'))
_MASKED = []  # native handles created here (kept alive for the life of the process)


def _masked_stream(d, cus_per_xcd):
  """An ExternalStream over a native stream restricted to cus_per_xcd CUs of each of the 8 XCDs (CU bit i = CU i in the device's enumeration,
  consecutive CUs alternate over the XCDs as the workgroup dispatcher deals them: bit i belongs to XCD i % 8)."""
  hip = ctypes.CDLL('libamdhip64.so')
  total = torch.cuda.get_device_properties(d).multi_processor_count
  words = (total + 31) // 32
  mask = (ctypes.c_uint32 * words)()
  for cu in range(total):
    if (cu // 8) < cus_per_xcd:
      mask[cu // 32] |= 1 << (cu % 32)
  handle = ctypes.c_void_p()
  with torch.cuda.device(d):
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), mask)
  if rc != 0 or not handle.value:
    raise RuntimeError(f'hipExtStreamCreateWithCUMask failed with code {rc}')
  _MASKED.append(handle)
  return torch.cuda.ExternalStream(handle.value, device=torch.device('cuda', d))


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
    if role == 'side' and 0 < _SIDE_CUS < 32:
      st = _STREAMS[key] = _masked_stream(d, _SIDE_CUS)
      return st
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
