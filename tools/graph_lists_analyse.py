#!/usr/bin/env python
"""Off-line analysis of tools/graph_lists.py's dump: emulates the depth-first run-list construction of the HIP runtime's graph executor
(each list = one internal stream whose nodes run strictly in list order; cross-list edges become event waits) on the captured training step
and prints the lists, so that serialisations the capture never asked for become visible."""
import gzip
import re
import sys
from collections import defaultdict

path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/step_graph.dot.gz'
names, edges = {}, defaultdict(list)
parents = defaultdict(list)
with gzip.open(path, 'rt') as f:
  text = f.read()
for m in re.finditer(r'\{ID \| (\d+) \| ([^\\]+)\\<\\<\\<\((\d+),(\d+),(\d+)\)', text):
  names[int(m.group(1))] = (m.group(2), int(m.group(3)) * int(m.group(4)) * int(m.group(5)))
for m in re.finditer(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', text):
  a, b = int(m.group(1)), int(m.group(2))
  edges[a].append(b)
  parents[b].append(a)
n = max(names) + 1
print(n, 'nodes', sum(len(v) for v in edges.values()), 'edges; roots:', [i for i in range(n) if not parents[i]][:10])

sys.setrecursionlimit(100000)
visited = [False] * n
lists = []
single = []


def util(v):
  visited[v] = True
  single.append(v)
  for a in edges[v]:
    if not visited[a]:
      util(a)
  if single:
    lists.append(list(single))
    single.clear()


for v in range(n):
  if not visited[v]:
    util(v)


def ranges(ids):
  out, s, p = [], ids[0], ids[0]
  for i in ids[1:]:
    if i != p + 1:
      out.append((s, p))
      s = i
    p = i
  out.append((s, p))
  return out


def short(i):
  nm = names[i][0]
  m = re.match(r'_Z\d*(\w+?)(?:I|P|E|v|RK|\d)', nm)
  return (m.group(1) if m else nm)[:28]


where = {}
for li, l in enumerate(lists):
  for pos, v in enumerate(l):
    where[v] = (li, pos)
print(len(lists), 'lists (internal streams)')
for li, l in enumerate(lists):
  r = ranges(l)
  print(f'list {li}: {len(l)} nodes; id ranges in list order: ' + ' '.join(f'{a}-{b}' if a != b else str(a) for a, b in r[:24]) + (' ...' if len(r) > 24 else ''))
# cross-list waits
cross = [(a, b) for a in edges for b in edges[a] if where[a][0] != where[b][0]]
print(len(cross), 'cross-list edges')
if '-v' in sys.argv:
  for a, b in cross:
    print(f'  {a} ({short(a)}, list {where[a][0]}) -> {b} ({short(b)}, list {where[b][0]})')
