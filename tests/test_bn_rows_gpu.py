"""GPU parity tests of the round-6 BatchNorm path (csrc/bn_rows_kernels.hip, normalise-on-load in conv3x3_halo.hip / wgrad3x3_halo.hip):
the statistics step runs in the prologue of the pass that consumes it, and conv1 / conv2 outputs of a RegNet bottleneck (timm Bottleneck:
conv -> BN(train) -> ReLU, oracle/timm_regnet.py) exist only as (raw convolution output, statistics).  Every entry point against plain
PyTorch fp32 on the CPU and against the launches of rounds 1-5 it replaces."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_ops_gpu import DEV, DTYPES, check, dev, nchw, nhwc, rnd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
  if not torch.cuda.is_available():
    pytest.skip('needs a GPU')
  from carla_garage_amd import ops as o
  return o


def _conv_raw(ops, x, w, dtype, k, G, stride=1):
  """raw conv output + its statistics rows (stored, one row per M-tile)"""
  B, Cin, H, W = x.shape
  Cout = w.shape[0]
  pad = k // 2
  Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
  xd = dev(nhwc(x), dtype)
  wp = ops.pack_conv_weight(dev(w), dtype, G=G)
  raw = torch.empty((B, Ho, Wo, Cout), device=DEV, dtype=dtype)
  geo = dict(B=B, Hs=H, Ws=W, Cs=Cin, Hd=Ho, Wd=Wo, Cd=Cout, R=k, S=k, stride=stride, pad=pad, G=G)
  nrows = ops.conv_gemm(xd, wp, raw, stats_rows_query=True, **geo)
  rows = torch.full((nrows * 2 * Cout,), float('nan'), device=DEV)  # stored, not accumulated: garbage in the buffer must not matter
  n2, _ = ops.conv_gemm(xd, wp, raw, stats_store=rows, **geo)
  assert n2 == nrows
  return raw, rows, nrows, geo, xd, wp


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', [(3, 12, 20, 40, 72, 1, 1), (2, 16, 64, 576, 576, 1, 1), (1, 8, 32, 576, 1512, 1, 1), (2, 10, 40, 48, 48, 3, 2), (4, 9, 11, 24, 216, 1, 1),
                                  (2, 7, 9, 16, 32, 1, 1)],
                         ids=['pw72', 's3_576', 's4_1512', 'g3x3_halo', 'pw216', 'c32'])
@pytest.mark.parametrize('mode', ['res_relu', 'plain', 'relu_gate'])
def test_bn_apply_rows_finalizes_in_its_prologue(ops, dtype, case, mode):
  """conv (statistics rows STORED by the epilogue) -> tfpp_bn_apply_rows: BatchNorm(train) + ReLU / residual / squeeze-excite gate in one
  launch, running statistics updated once, scale / shift / saved statistics left for later kernels -- against F.batch_norm."""
  B, H, W, Cin, Cout, k, G = case
  x = rnd(B, Cin, H, W, dtype=dtype, seed=61)
  w = (rnd(Cout, Cin // G, k, k, seed=62) * (1.0 / math.sqrt(Cin // G * k * k))).to(dtype).float()
  gamma, beta = rnd(Cout, seed=63, lo=0.5, hi=1.5), rnd(Cout, seed=64)
  rm, rv = rnd(Cout, seed=65), rnd(Cout, seed=66, lo=0.5, hi=1.5)
  rm_ref, rv_ref = rm.clone(), rv.clone()
  conv = F.conv2d(x, w, None, 1, k // 2, 1, G)
  if dtype == torch.bfloat16:
    conv = conv.to(dtype).float()
  bn = F.batch_norm(conv, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
  res = rnd(B, Cout, H, W, dtype=dtype, seed=67)
  gate = rnd(B, Cout, seed=68, lo=0.0, hi=1.0)
  if mode == 'res_relu':
    want = F.relu(bn + res)
  elif mode == 'plain':
    want = bn
  else:
    want = F.relu(bn) * gate.view(B, Cout, 1, 1)
  raw, rows, nrows, _, _, _ = _conv_raw(ops, x, w, dtype, k, G)
  assert nrows <= ops.BN_ROWS_MAX
  scale, shift, sm, si = (torch.full((Cout,), float('nan'), device=DEV) for _ in range(4))
  rmd, rvd, nbt = dev(rm), dev(rv), torch.zeros((), device=DEV, dtype=torch.long)
  keep = [dev(gamma), dev(beta)]
  b = ops.bn_rows(Cout, scale, shift, partial=rows, nrows=nrows, count=B * H * W, gamma=keep[0], beta=keep[1], rm=rmd, rv=rvd, nbt=nbt, save_mean=sm,
                  save_invstd=si)
  if mode == 'res_relu':
    y = ops.bn_apply_rows(raw, b, res=dev(nhwc(res), dtype), relu_post=True)
  elif mode == 'plain':
    y = ops.bn_apply_rows(raw, b)
  else:
    y = ops.bn_apply_rows(raw, b, gate=dev(gate), rows_per_batch=H * W, relu_pre=True)
  check(f'bnrows.{mode}', nchw(y.float().cpu()), want, dtype, scale=3.0)
  check('bnrows.running_mean', rmd.cpu(), rm_ref, dtype, scale=1.0 if dtype == torch.bfloat16 else 5.0)
  check('bnrows.running_var', rvd.cpu(), rv_ref, dtype, scale=1.0 if dtype == torch.bfloat16 else 5.0)
  assert int(nbt.item()) == 1
  # what the prologue wrote equals the separate finalize launch of rounds 1-5 on the same rows
  s2, h2, m2, i2 = (torch.empty(Cout, device=DEV) for _ in range(4))
  ops.bn_finalize_partials(rows, nrows, keep[0], keep[1], None, None, None, s2, h2, m2, i2, B * H * W, clear=False)
  for nme, a, c in (('scale', scale, s2), ('shift', shift, h2), ('mean', sm, m2), ('invstd', si, i2)):
    torch.testing.assert_close(a, c, rtol=1e-6, atol=1e-7, msg=f'{nme} written by the prologue differs from bn_finalize_partials')
  # a later kernel of the pass reads scale / shift instead of the rows: same result
  b2 = ops.bn_rows(Cout, scale, shift)
  if mode == 'plain':
    assert torch.equal(ops.bn_apply_rows(raw, b2), y)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C,B,H,W', [(72, 3, 12, 20), (216, 2, 32, 24), (576, 12, 16, 64), (1512, 2, 8, 32), (32, 2, 10, 14)])
@pytest.mark.parametrize('mask', ['y', 'raw', 'none'])
def test_bn_backward_with_the_coefficient_step_in_the_prologue(ops, dtype, C, B, H, W, mask):
  """tfpp_bn_bwd_reduce_rows + tfpp_bn_bwd_apply_rows2 against autograd through relu(batch_norm(x) [+ res]); mask 'raw' recomputes the ReLU
  mask from the raw tensor (the normalised tensor does not exist), 'y' reads the forward output."""
  x = (rnd(B, C, H, W, dtype=dtype, seed=41) * 2.0 + 0.7).to(dtype).float()
  gamma, beta = rnd(C, seed=42, lo=0.5, hi=1.5), rnd(C, seed=43)
  xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
  bn = F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5)
  res = rnd(B, C, H, W, dtype=dtype, seed=46) if mask == 'y' else None
  resr = res.clone().requires_grad_(True) if res is not None else None
  want = bn if mask == 'none' else F.relu(bn + resr if resr is not None else bn)
  dy = rnd(B, C, H, W, dtype=dtype, seed=47)
  want.backward(dy)
  xd, dyd = dev(nhwc(x), dtype), dev(nhwc(dy), dtype)
  mean = x.mean((0, 2, 3))
  var = x.var((0, 2, 3), unbiased=False)
  invstd = 1.0 / torch.sqrt(var + 1e-5)
  scale, shift = gamma * invstd, beta - mean * gamma * invstd
  sm, si, sc, sh = dev(mean), dev(invstd), dev(scale), dev(shift)
  dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
  if mask == 'y':
    y = dev(nhwc(want.detach()), dtype)
    m = ops.MASK_Y
  else:
    y = None
    m = ops.MASK_RAW if mask == 'raw' else ops.MASK_NONE
  partial, nrows = ops.bn_bwd_reduce_rows(dyd, y, xd, sc, sh, sm, si, m)
  assert 1 <= nrows <= ops.BN_ROWS_MAX // 2
  dx, dres = ops.bn_bwd_apply_rows2(dyd, y, xd, sc, sh, dev(gamma), sm, si, partial, nrows, dgamma, dbeta, m, want_dres=mask == 'y')
  tag = f'bnrows_bwd{C}.{mask}'
  check(tag + '.dx', nchw(dx.float().cpu()), xr.grad, dtype, scale=5.0)
  check(tag + '.dgamma', dgamma.cpu(), gr.grad, dtype, scale=5.0)
  check(tag + '.dbeta', dbeta.cpu(), br.grad, dtype, scale=5.0)
  if mask == 'y':
    check(tag + '.dres', nchw(dres.float().cpu()), resr.grad, dtype, scale=5.0)
    # the rounds 1-5 launches on the same tensors
    dg2, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dx2, dres2 = ops.bn_bwd(dyd, y, xd, dev(gamma), sm, si, None, dg2, db2, relu_mask=True, want_dres=True)
    assert torch.equal(dres2, dres)
    check(tag + '.dx_vs_r5', dx.float().cpu(), dx2.float().cpu(), dtype, scale=0.5)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C,RD,B,H,W', [(72, 18, 3, 16, 20), (216, 54, 2, 32, 32), (576, 144, 12, 16, 64), (1512, 144, 2, 8, 32)])
def test_squeeze_excite_around_a_tensor_that_is_never_normalised_in_memory(ops, dtype, C, RD, B, H, W):
  """mean_hw_bn (finalize prologue) -> gate MLP -> bn_apply_rows(relu_pre, gate) forward; se_dgate_bn, se_gate_bwd, se_bwd_apply_bn backward,
  against autograd through relu(batch_norm(raw)) * sigmoid(fc2(relu(fc1(mean)))) (timm SEModule behind conv2 + BN + ReLU)."""
  raw = (rnd(B, C, H, W, dtype=dtype, seed=11) * 1.5 + 0.2).to(dtype).float()
  gamma, beta = rnd(C, seed=12, lo=0.5, hi=1.5), rnd(C, seed=13)
  w1, b1, w2, b2 = rnd(RD, C, seed=14) * 0.3, rnd(RD, seed=15), rnd(C, RD, seed=16) * 0.5, rnd(C, seed=17)
  rawr = raw.clone().requires_grad_(True)
  w1r, b1r, w2r, b2r = (t.clone().requires_grad_(True) for t in (w1, b1, w2, b2))
  y2 = F.relu(F.batch_norm(rawr, None, None, gamma, beta, True, 0.1, 1e-5))
  y2.retain_grad()
  pool_ref = y2.mean((2, 3))
  gate_ref = torch.sigmoid(F.linear(F.relu(F.linear(pool_ref, w1r, b1r)), w2r, b2r))
  a2_ref = y2 * gate_ref.view(B, C, 1, 1)
  d_a2 = rnd(B, C, H, W, dtype=dtype, seed=18)
  a2_ref.backward(d_a2)
  # statistics rows as a convolution epilogue would leave them: a few row blocks of (sum, sum of squares)
  rd = dev(nhwc(raw), dtype)
  flat = rd.float().view(-1, C)
  nrows = 7
  chunks = flat.chunk(nrows, 0)
  rows = torch.stack([torch.cat([c.sum(0), (c * c).sum(0)]) for c in chunks]).contiguous().view(-1)
  scale, shift, sm, si = (torch.full((C,), float('nan'), device=DEV) for _ in range(4))
  keep = [dev(gamma), dev(beta)]
  bn = ops.bn_rows(C, scale, shift, partial=rows, nrows=len(chunks), count=B * H * W, gamma=keep[0], beta=keep[1], save_mean=sm, save_invstd=si)
  pool = ops.mean_hw_bn(rd, bn)
  check('se_bn.pool', pool.cpu(), pool_ref, dtype)
  hidden, gate = ops.se_gate_fwd(pool, dev(w1), dev(b1), dev(w2), dev(b2))
  check('se_bn.gate', gate.cpu(), gate_ref, dtype)
  a2 = ops.bn_apply_rows(rd, ops.bn_rows(C, scale, shift), gate=gate, rows_per_batch=H * W, relu_pre=True)
  check('se_bn.a2', nchw(a2.float().cpu()), a2_ref, dtype, scale=2.0)
  dyd = dev(nhwc(d_a2), dtype)
  dgate_g = ops.se_dgate(dyd, a2)  # = dgate * gate, on the gated tensor as stored (tfpp_se_gate_bwd_premul)
  check('se_bn.dgate_g', dgate_g.cpu(), (d_a2 * a2_ref.detach()).sum((2, 3)), dtype, scale=3.0)
  grads = [torch.zeros_like(dev(t)) for t in (w1, b1, w2, b2)]
  dpool = ops.se_gate_bwd(dgate_g, gate, hidden, pool, dev(w1), dev(w2), *grads, premul=True)
  for nme, gg, pp in zip(('dw1', 'db1', 'dw2', 'db2'), grads, (w1r, b1r, w2r, b2r)):
    check('se_bn.' + nme, gg.cpu(), pp.grad, dtype, scale=4.0)
  d_y2, partial, nr = ops.se_bwd_apply_bn(dyd, gate, dpool, rd, scale, shift, sm, si)
  assert nr <= ops.BN_ROWS_MAX // 2 + B
  check('se_bn.d_y2', nchw(d_y2.float().cpu()), y2.grad, dtype, scale=3.0)
  dgamma, dbeta = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
  d_raw, _ = ops.bn_bwd_apply_rows2(d_y2, None, rd, scale, shift, keep[0], sm, si, partial, nr, dgamma, dbeta, ops.MASK_RAW)
  check('se_bn.d_raw', nchw(d_raw.float().cpu()), rawr.grad, dtype, scale=6.0)


@pytest.mark.parametrize('case', [(3, 16, 64, 24, 24, 3, 1), (2, 32, 32, 72, 72, 3, 1), (2, 8, 32, 1512 // 7, 1512 // 7, 9, 1), (2, 32, 64, 72, 72, 3, 2), (12, 16, 64, 576, 576, 24, 1)],
                         ids=['g24x3', 'g24x3_32', 's4_g24x9', 'stride2', 's3_full'])
def test_conv3x3_normalises_its_source_while_staging_it(ops, case):
  """conv1 -> [BN(train) + ReLU on load] -> grouped 3x3 conv2 (tfpp_conv_params.in_bn) against the same conv2 on the materialised tensor;
  the weight gradient of conv2 with x normalised on load (tfpp_wgrad_params.x_scale) against the one on the materialised tensor."""
  dtype = torch.bfloat16
  B, H, W, Cin1, C, G, stride = case
  x = rnd(B, Cin1, H, W, dtype=dtype, seed=21)
  w1 = (rnd(C, Cin1, 1, 1, seed=22) * (1.0 / math.sqrt(Cin1))).to(dtype).float()
  w2 = (rnd(C, C // G, 3, 3, seed=23) * (1.0 / math.sqrt(C // G * 9))).to(dtype).float()
  gamma, beta = rnd(C, seed=24, lo=0.5, hi=1.5), rnd(C, seed=25)
  raw1, rows, nrows, _, _, _ = _conv_raw(ops, x, w1, dtype, 1, 1)
  scale, shift, sm, si = (torch.full((C,), float('nan'), device=DEV) for _ in range(4))
  rmd, rvd, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), device=DEV, dtype=torch.long)
  keep = [dev(gamma), dev(beta)]
  Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
  geo = dict(B=B, Hs=H, Ws=W, Cs=C, Hd=Ho, Wd=Wo, Cd=C, R=3, S=3, stride=stride, pad=1, G=G)
  wp2 = ops.pack_conv_weight(dev(w2), dtype, G=G)
  out_bn = torch.empty((B, Ho, Wo, C), device=DEV, dtype=dtype)
  assert ops.conv_gemm(raw1, wp2, out_bn, in_bn_query=True, **geo)
  b = ops.bn_rows(C, scale, shift, partial=rows, nrows=nrows, count=B * H * W, gamma=keep[0], beta=keep[1], rm=rmd, rv=rvd, nbt=nbt, save_mean=sm, save_invstd=si)
  n2 = ops.conv_gemm(raw1, wp2, out_bn, stats_rows_query=True, **geo)
  rows2 = torch.full((n2 * 2 * C,), float('nan'), device=DEV)
  ops.conv_gemm(raw1, wp2, out_bn, in_bn=b, in_relu=True, stats_store=rows2, **geo)
  assert int(nbt.item()) == 1 and torch.isfinite(scale).all() and torch.isfinite(si).all()
  # reference: materialise y1 = relu(BN(raw1)) with the same statistics, then the plain kernel
  y1 = ops.bn_apply_rows(raw1, ops.bn_rows(C, scale, shift), relu_pre=True)
  out_ref = torch.empty_like(out_bn)
  rows_ref = torch.full((n2 * 2 * C,), float('nan'), device=DEV)
  ops.conv_gemm(y1, wp2, out_ref, stats_store=rows_ref, **geo)
  assert torch.equal(out_bn, out_ref), float((out_bn.float() - out_ref.float()).abs().max())
  assert torch.equal(rows2, rows_ref)
  want = F.conv2d(nchw(y1.float().cpu()), w2, None, stride, 1, 1, G)
  check('halo_in_bn.fwd', nchw(out_bn.float().cpu()), want, dtype)
  if stride != 1:
    return
  # weight gradient of conv2: dW = dY^T y1 with y1 rebuilt from raw1 while the halo is staged
  dy = dev(nhwc(rnd(B, C, Ho, Wo, dtype=dtype, seed=26)), dtype)
  dw_ref, dw_bn = torch.zeros(C, C // G, 3, 3, device=DEV), torch.zeros(C, C // G, 3, 3, device=DEV)
  wg = dict(B=B, Hs=H, Ws=W, Cs=C, Hd=Ho, Wd=Wo, Cd=C, R=3, S=3, stride=1, pad=1, G=G, ks_g=C // G, n_g=C // G, c_real=C // G)
  assert ops.conv_wgrad_x_bn_ok(dy, raw1, dw_bn, **wg)
  ops.conv_wgrad(dy, y1, dw_ref, **wg)
  ops.conv_wgrad(dy, raw1, dw_bn, x_scale=scale, x_shift=shift, x_relu=True, **wg)
  assert torch.equal(dw_bn, dw_ref), float((dw_bn - dw_ref).abs().max())
