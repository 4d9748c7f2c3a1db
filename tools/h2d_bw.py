"""raw H2D rates on this box: pinned vs pageable, one 43 MB buffer vs 20 pieces, idle GPU vs under a running kernel stream"""
import time
import torch
n = 43 * 1024 * 1024
dev = torch.empty(n, dtype=torch.uint8, device='cuda')
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pag = torch.empty(n, dtype=torch.uint8)
print('is_pinned', pin.is_pinned())
cs = torch.cuda.Stream()
def t(fn, it=10):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(it): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / it * 1e3
print('pinned one buffer ms', t(lambda: dev.copy_(pin, non_blocking=True)))
print('pageable one buffer ms', t(lambda: dev.copy_(pag)))
def pieces():
  k = n // 20
  for i in range(20): dev[i * k:(i + 1) * k].copy_(pin[i * k:(i + 1) * k], non_blocking=True)
print('pinned 20 pieces ms', t(pieces))
def on_cs():
  with torch.cuda.stream(cs): dev.copy_(pin, non_blocking=True)
print('pinned copy stream ms', t(on_cs))
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
def busy():
  for _ in range(10): a @ a
print('10 gemms ms', t(busy))
def both():
  on_cs(); busy()
print('gemms + copy on copy stream ms', t(both))
