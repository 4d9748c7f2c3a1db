#!/usr/bin/env python
"""How does the HIP runtime lay the captured training step out on streams?  Captures the bs = 12 step with keep_graph, reads nodes and edges
back through hipGraphGetNodes / hipGraphGetEdges (and hipGraphDebugDotPrint for the kernel names), and writes them to
gpurun_out/step_graph_{nodes,edges}.txt + step_graph.dot.gz for off-line analysis (tools/graph_lists_analyse.py)."""
import ctypes
import gzip
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from carla_garage_amd import graph as G  # noqa: E402
from carla_garage_amd.config import GlobalConfig  # noqa: E402
from carla_garage_amd.model import LidarCenterNet  # noqa: E402
from carla_garage_amd.trainer import Trainer  # noqa: E402


def main():
  out = os.path.join(ROOT, 'gpurun_out')
  os.makedirs(out, exist_ok=True)
  torch.manual_seed(0)
  cfg = GlobalConfig(tfpp_dtype='bf16')
  dev = torch.device('cuda:0')
  model = LidarCenterNet(cfg).to(dev).train()
  tr = Trainer(model, lr=1e-5)
  batch = bench.synthetic_batch(12, cfg, dev, 1234)
  for _ in range(2):
    tr.train_step(batch)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph(keep_graph=True)
  st = G.capture_stream(tr.eng.device)
  with torch.cuda.graph(g, stream=st, capture_error_mode=G.CAPTURE_MODE):
    tr._step_body(batch)
  raw = g.raw_cuda_graph()
  print('raw graph handle', hex(raw))
  hip = ctypes.CDLL('libamdhip64.so')
  n = ctypes.c_size_t(0)
  rc = hip.hipGraphGetNodes(ctypes.c_void_p(raw), None, ctypes.byref(n))
  print('hipGraphGetNodes rc', rc, 'nodes', n.value)
  nodes = (ctypes.c_void_p * n.value)()
  hip.hipGraphGetNodes(ctypes.c_void_p(raw), nodes, ctypes.byref(n))
  ne = ctypes.c_size_t(0)
  rc = hip.hipGraphGetEdges(ctypes.c_void_p(raw), None, None, ctypes.byref(ne))
  print('hipGraphGetEdges rc', rc, 'edges', ne.value)
  fr, to = (ctypes.c_void_p * ne.value)(), (ctypes.c_void_p * ne.value)()
  hip.hipGraphGetEdges(ctypes.c_void_p(raw), fr, to, ctypes.byref(ne))
  idx = {int(nodes[i] or 0): i for i in range(n.value)}
  with open(os.path.join(out, 'step_graph_edges.txt'), 'w') as f:
    for i in range(ne.value):
      f.write(f'{idx[int(fr[i])]} {idx[int(to[i])]}\n')
  # node types (+ kernel names where the runtime gives them)
  hip.hipKernelNameRefByPtr.restype = ctypes.c_char_p
  with open(os.path.join(out, 'step_graph_nodes.txt'), 'w') as f:
    for i in range(n.value):
      t = ctypes.c_int(-1)
      hip.hipGraphNodeGetType(nodes[i], ctypes.byref(t))
      name = ''
      if t.value == 0:  # hipGraphNodeTypeKernel
        class KP(ctypes.Structure):
          _fields_ = [('blockDim', ctypes.c_uint * 3), ('extra', ctypes.c_void_p), ('func', ctypes.c_void_p), ('gridDim', ctypes.c_uint * 3),
                      ('kernelParams', ctypes.c_void_p), ('sharedMemBytes', ctypes.c_uint)]
        kp = KP()
        if hip.hipGraphKernelNodeGetParams(nodes[i], ctypes.byref(kp)) == 0 and kp.func:
          try:
            nm = hip.hipKernelNameRefByPtr(ctypes.c_void_p(kp.func), None)
            name = (nm or b'').decode()[:100] + f' grid {kp.gridDim[0]}x{kp.gridDim[1]}x{kp.gridDim[2]}'
          except Exception as e:  # noqa: BLE001
            name = f'? {e}'
      f.write(f'{i} {t.value} {name}\n')
  dot = os.path.join(out, 'step_graph.dot')
  rc = hip.hipGraphDebugDotPrint(ctypes.c_void_p(raw), dot.encode(), ctypes.c_uint(1))
  print('hipGraphDebugDotPrint rc', rc, os.path.exists(dot) and os.path.getsize(dot))
  if os.path.exists(dot):
    with open(dot, 'rb') as a, gzip.open(dot + '.gz', 'wb') as b:
      shutil.copyfileobj(a, b)
    os.remove(dot)
  g.replay()
  torch.cuda.synchronize()
  print('replayed')


if __name__ == '__main__':
  main()
