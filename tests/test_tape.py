"""Host logic of the reverse-mode tape (engine.Tape) on CPU: the accumulation kernels are replaced by torch ops, so the
bookkeeping -- fan-out accumulation, the frozen-gradient rule, take_pending, the two-segment backward used for the
overlapped gradient exchange -- is tested without a GPU."""
import pytest
import torch

from carla_garage_amd import engine as E


@pytest.fixture()
def cpu_ops(monkeypatch):
  calls = {'axpy': 0, 'add': 0}

  def axpy(x, y, a=1.0):
    calls['axpy'] += 1
    y.add_(x, alpha=a)
    return y

  def add_dropout(a, b, p_drop=0.0, seed=0, out=None):
    calls['add'] += 1
    return a + b

  monkeypatch.setattr(E.ops, 'axpy', axpy)
  monkeypatch.setattr(E.ops, 'add_dropout', add_dropout)
  monkeypatch.setattr(E.ops, 'cast', lambda t, dt: t.to(dt))
  return calls


def _toy(tape, x):
  """y = 3*a + b with a = 2*x, b = a*a (fan-out of a), z = y + x (fan-out of x); returns (z, grads dict filled by backward)."""
  got = {}
  a = 2 * x
  tape.record([a], [x], lambda d: 2 * d)
  b = a * a
  tape.record([b], [a], lambda d: 2 * a * d)
  y = 3 * a + b
  tape.record([y], [a, b], lambda d: (3 * d, d.clone()))
  z = y + x
  tape.record([z], [y, x], lambda d: (d, d))  # the same gradient object for two inputs
  tape.record([x], [], lambda d: got.__setitem__('x', d.clone()) or ())
  # the leaf node is recorded last but x is produced first: move it to the front so it is visited last in reverse order
  tape.nodes.insert(0, tape.nodes.pop())
  return z, got


def _want(x):
  xr = x.clone().requires_grad_(True)
  a = 2 * xr
  z = 3 * a + a * a + xr
  z.backward(torch.ones_like(z))
  return xr.grad


def test_single_segment_backward_accumulates_fan_out(cpu_ops):
  x = torch.arange(1.0, 7.0)
  t = E.Tape()
  z, got = _toy(t, x)
  t.backward([(z, torch.ones_like(z))])
  assert torch.allclose(got['x'], _want(x))
  assert E.Tape.current is None and t.nodes == [] and t._grads is None
  assert cpu_ops['axpy'] + cpu_ops['add'] >= 2  # a and x each receive two contributions


def test_shared_gradient_object_is_not_mutated(cpu_ops):
  """z = y + x hands ONE tensor to both inputs.  While it is still pending for y, another contribution to x arrives: it must be
  added out of place, otherwise y's gradient would change under its feet."""
  x = torch.arange(1.0, 4.0)
  t = E.Tape()
  seen = {}
  y = x * 1.0
  t.record([y], [x], lambda d: seen.__setitem__('dy', d.clone()) or d * 5)
  u = x * 2.0
  t.record([u], [x], lambda d: d * 2)
  z = y + x
  t.record([z], [y, x], lambda d: (d, d))
  t.nodes.insert(0, ([x], [], lambda d: seen.__setitem__('dx', d.clone()) or (), 0))
  # reverse order: z (shared object pending for y and x), u (second contribution to x while y's is still pending), y, leaf
  t.backward([(z, torch.ones_like(z)), (u, torch.ones_like(u))])
  assert torch.equal(seen['dy'], torch.ones(3))                 # untouched by the accumulation into x
  assert torch.equal(seen['dx'], torch.full((3,), 1.0 + 2.0 + 5.0))
  assert cpu_ops['add'] == 1                                    # the shared object was not mutated ...
  assert cpu_ops['axpy'] == 1                                   # ... the private sum that replaced it was


def test_frozen_gradients_are_accumulated_out_of_place_and_take_pending_skips_them(cpu_ops):
  x = torch.arange(1.0, 4.0)
  t = E.Tape()
  y1, y2 = x * 2, x * 3
  g1 = torch.ones(3)
  seen = {}

  def bwd1(d):
    t.freeze(d)  # handed to the weight-gradient lane
    seen['frozen'] = d
    return d

  t.record([y1], [x], bwd1)
  t.record([y2], [x], lambda d: (seen.__setitem__('pend', t.take_pending(x, d)), d * 3)[1])
  t.nodes.insert(0, ([x], [], lambda d: seen.__setitem__('dx', d.clone()) or (), 0))
  t.backward([(y1, g1), (y2, torch.ones(3))])
  # reverse order: y2's node runs first (no pending gradient for x yet -> None), then y1's freezes its gradient object
  assert seen['pend'] is None
  assert torch.equal(seen['dx'], torch.full((3,), 4.0))
  assert torch.equal(seen['frozen'], torch.ones(3))  # the frozen tensor itself was never written


def test_take_pending_consumes_an_unshared_gradient(cpu_ops):
  x = torch.arange(1.0, 4.0)
  t = E.Tape()
  y1, y2 = x * 2, x * 3
  seen = {}
  t.record([y2], [x], lambda d: d * 3)

  def bwd1(d):
    pend = t.take_pending(x, d)  # the contribution of y2 is already pending: fuse it into this node's own result
    seen['pend'] = None if pend is None else pend.clone()
    return d * 2 + pend

  t.record([y1], [x], bwd1)
  t.nodes.insert(0, ([x], [], lambda d: seen.__setitem__('dx', d.clone()) or (), 0))
  t.nodes[1], t.nodes[2] = t.nodes[2], t.nodes[1]  # visit y2 first, then y1
  t.backward([(y1, torch.ones(3)), (y2, torch.ones(3))])
  assert torch.equal(seen['pend'], torch.full((3,), 3.0)) and torch.equal(seen['dx'], torch.full((3,), 5.0))
  assert cpu_ops['axpy'] == 0 and cpu_ops['add'] == 0  # no separate accumulation pass


def test_relane_changes_only_the_lane_backward_runs_a_node_on(cpu_ops):
  """Tape.relane (engine: backward of LiDAR stage 1 runs on lane 0): the nodes keep outputs, inputs and closure; on one stream (CPU) the
  gradients are those of the unchanged tape."""
  x = torch.arange(6, dtype=torch.float32).view(2, 3) / 7
  tape = E.Tape()
  z, got = _toy(tape, x)
  before = [(o, i, f) for o, i, f, _ in tape.nodes]
  tape.nodes = [(o, i, f, 1) for o, i, f, _ in tape.nodes]     # as if recorded on a branch lane
  tape.relane(1, 3, 0)
  assert [n[3] for n in tape.nodes] == [1, 0, 0, 1, 1]
  assert all(a[0] is b[0] and a[1] is b[1] and a[2] is b[2] for a, b in zip(before, tape.nodes))
  tape.relane(3, 99, 0)                                        # a range past the end is clipped
  assert [n[3] for n in tape.nodes] == [1, 0, 0, 0, 0]
  tape.backward([(z, torch.ones_like(z))])
  torch.testing.assert_close(got['x'], _want(x))
