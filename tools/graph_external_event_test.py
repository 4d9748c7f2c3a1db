#!/usr/bin/env python
"""How can a stream that is NOT part of a captured hipGraph wait for a point INSIDE the replay in flight?

The N > 1 training step keeps ONE captured graph and lets the RCCL stream start the all-reduce of a gradient bucket as soon as the bucket is
complete, i.e. in the middle of the replay.  Three candidates, each tested the same way: the graph runs
    spin(~2 ms) -> flag = epoch -> [sync point] -> spin(~2 ms)
a side stream waits for the sync point and copies the flag.  Correct = the copy sees THIS replay's epoch and completes well BEFORE the graph's
tail does (so the wait really is on the point, not on the whole graph, and not on an earlier replay).

  A  torch.cuda.Event(external=True)                                   -- PyTorch refuses it on ROCm ("External events are disallowed in rocm")
  B  hipEventRecordWithFlags(ev, stream, hipEventRecordExternal) through ctypes, hipStreamWaitEvent on the side stream
  C  a counter in signal memory (hipExtMallocWithFlags(hipMallocSignalMemory)) incremented by a kernel node of the graph (tfpp_inc_u64),
     hipStreamWaitValue64(side, counter, replay number, hipStreamWaitValueGte) on the side stream -- the command processor polls, no CU is held

  D  the library's own pair: tfpp_inc_u64 (one-thread kernel node, device-scope atomic) + tfpp_signal_wait (one-wave polling kernel on the
     side stream)

Prints one line per replay and a verdict per candidate (carla_garage_amd/buckets.py uses D; C -- plain device memory works, signal memory
is not available -- would hold no CU but is a BETA API reached through ctypes)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def hip():
  for name in ('libamdhip64.so', '/opt/rocm/lib/libamdhip64.so'):
    try:
      return ctypes.CDLL(name)
    except OSError:
      continue
  raise RuntimeError('libamdhip64.so not found')


def run(name, record, wait, replays=8):
  dev = torch.device('cuda')
  flag = torch.zeros(1, device=dev, dtype=torch.int64)
  epoch = torch.zeros(1, device=dev, dtype=torch.int64)
  seen = torch.zeros(1, device=dev, dtype=torch.int64)
  side = torch.cuda.Stream(dev)
  spin = int(2e-3 * 2.0e9)
  torch.cuda._sleep(1000)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  cap = torch.cuda.Stream(dev)
  try:
    with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
      torch.cuda._sleep(spin)
      flag.copy_(epoch)
      record(torch.cuda.current_stream())
      torch.cuda._sleep(spin)
  except Exception as e:  # pylint: disable=broad-except
    print(f'{name}: capture failed: {type(e).__name__}: {e}', flush=True)
    return False
  ok = True
  for it in range(1, replays + 1):
    epoch.fill_(it)
    seen.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    try:
      wait(side, it)
    except Exception as e:  # pylint: disable=broad-except
      print(f'{name}: wait failed: {type(e).__name__}: {e}', flush=True)
      torch.cuda.synchronize()
      return False
    with torch.cuda.stream(side):
      seen.copy_(flag)
      done_side = torch.cuda.Event()
      done_side.record(side)
    done_side.synchronize()
    t_side = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    v = int(seen.item())
    good = (v == it) and (t_side < 0.8 * t_all or it == 1)  # (the first replay loads code objects: its timing says nothing)
    ok &= good
    print(f'{name} replay {it}: side stream saw epoch {v} after {t_side * 1e3:.2f} ms, graph finished after {t_all * 1e3:.2f} ms -> {"ok" if good else "WRONG"}', flush=True)
  print(f'{name}:', 'OK' if ok else 'BROKEN', flush=True)
  return ok


def main():
  if len(sys.argv) < 2:  # every candidate in its own process: a failed capture leaves a sticky HIP error behind
    import subprocess
    for c in 'ABCD':
      r = subprocess.run([sys.executable, os.path.abspath(__file__), c], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150, check=False)
      text = r.stdout.decode()
      print('\n'.join(l for l in text.splitlines() if 'amdgpu.ids' not in l)[-1500:], flush=True)
    return
  only = sys.argv[1]
  H = hip()
  results = {}
  # ---- A: PyTorch's own external event
  if only == 'A':
   try:
    evA = torch.cuda.Event(external=True)
    results['A torch external event'] = run('A', lambda st: evA.record(st), lambda side, it: side.wait_event(evA))
   except Exception as e:  # pylint: disable=broad-except
    print(f'A: {type(e).__name__}: {e}', flush=True)
    results['A torch external event'] = False
  # ---- B: raw HIP external event
  evB = ctypes.c_void_p()
  rc = H.hipEventCreateWithFlags(ctypes.byref(evB), ctypes.c_uint(0x0)) if only == 'B' else -1  # default flags
  if rc == 0:
    def recB(st):
      r = H.hipEventRecordWithFlags(evB, ctypes.c_void_p(st.cuda_stream), ctypes.c_uint(0x1))  # hipEventRecordExternal
      if r != 0:
        raise RuntimeError(f'hipEventRecordWithFlags -> {r}')

    def waitB(side, it):
      r = H.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), evB, ctypes.c_uint(0))
      if r != 0:
        raise RuntimeError(f'hipStreamWaitEvent -> {r}')
    results['B raw hip external event'] = run('B', recB, waitB)
  elif only == 'B':
    print(f'B: hipEventCreateWithFlags -> {rc}', flush=True)
    results['B raw hip external event'] = False
  # ---- C: counter in signal memory + hipStreamWaitValue64
  attr = ctypes.c_int(0)
  H.hipDeviceGetAttribute(ctypes.byref(attr), ctypes.c_int(0), ctypes.c_int(0))  # (attribute ids differ between releases: informative only)
  sig = ctypes.c_void_p()
  rc = H.hipExtMallocWithFlags(ctypes.byref(sig), ctypes.c_size_t(8), ctypes.c_uint(0x2)) if only == 'C' else -1  # hipMallocSignalMemory (8 bytes exactly)
  if rc != 0 and only == 'C':
    print(f'C: hipExtMallocWithFlags(8, hipMallocSignalMemory) -> {rc}; trying plain device memory', flush=True)
    rc = H.hipMalloc(ctypes.byref(sig), ctypes.c_size_t(64))
  if rc == 0:
    H.hipMemset(sig, 0, ctypes.c_size_t(8))
    from carla_garage_amd import _lib
    inc = _lib.lib.raw('tfpp_inc_u64')
    H.hipStreamWaitValue64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint64]

    def recC(st):
      r = inc(sig.value, st.cuda_stream)
      if r != 0:
        raise RuntimeError(f'tfpp_inc_u64 -> {r}')

    def waitC(side, it):
      r = H.hipStreamWaitValue64(ctypes.c_void_p(side.cuda_stream), sig, ctypes.c_uint64(it), ctypes.c_uint(0), ctypes.c_uint64(0xFFFFFFFFFFFFFFFF))
      if r != 0:
        raise RuntimeError(f'hipStreamWaitValue64 -> {r}')
    results['C signal counter + hipStreamWaitValue64'] = run('C', recC, waitC)
  elif only == 'C':
    print(f'C: hipExtMallocWithFlags(hipMallocSignalMemory) -> {rc}', flush=True)
    results['C signal counter + hipStreamWaitValue64'] = False
  # ---- D: the library's own device-side signal (carla_garage_amd/buckets.py): tfpp_inc_u64 node + tfpp_signal_wait polling kernel
  if only != 'D':
    for k, v in results.items():
      print('RESULT', k, 'OK' if v else 'not usable', flush=True)
    return
  from carla_garage_amd import _lib as L2
  from carla_garage_amd import ops
  sigD = ops.zeros(1, torch.int64, torch.device('cuda'))
  tmo = ops.zeros(1, torch.int32, torch.device('cuda'))
  addD, waitD = L2.lib.raw('tfpp_inc_u64'), L2.lib.raw('tfpp_signal_wait')

  def recD(st):
    if addD(sigD.data_ptr(), st.cuda_stream) != 0:
      raise RuntimeError('tfpp_inc_u64 failed')

  def wD(side, it):
    if waitD(sigD.data_ptr(), it, 2000, tmo.data_ptr(), side.cuda_stream) != 0:
      raise RuntimeError('tfpp_signal_wait failed')
  results['D device counter + polling kernel (tfpp_signal_*)'] = run('D', recD, wD) and int(tmo.item()) == 0
  for k, v in results.items():
    print('RESULT', k, 'OK' if v else 'not usable', flush=True)
  sys.exit(0)


if __name__ == '__main__':
  main()
