"""Tensor-level wrappers over the C ABI (include/tfpp.h).  PyTorch is used for device memory and streams only:
every function takes CUDA(HIP) tensors, passes raw device pointers + the current stream to ``libtfpp_hip.so``
and returns/fills tensors allocated with ``torch.empty``.  No ATen compute kernels run here."""
import ctypes

import torch

from ._lib import (lib, ConvParams, WgradParams, BgemmParams, PackDesc, AttnParams, BnRows, F32, BF16, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_GELU,
                   ACT_TANH)

import os as _os

PROFILE_SHAPES = bool(int(_os.environ.get('TFPP_PROFILE_SHAPES', '0')))

__all__ = ['F32', 'BF16', 'ACT_NONE', 'ACT_RELU', 'ACT_SIGMOID', 'ACT_GELU', 'ACT_TANH']


def dt(t):
  if t.dtype == torch.float32:
    return F32
  if t.dtype == torch.bfloat16:
    return BF16
  raise TypeError(f'unsupported dtype {t.dtype}')


def vec(dtype):
  return 4 if dtype == torch.float32 else 8


def pad_to(n, m):
  return (n + m - 1) // m * m


def stream():
  return torch.cuda.current_stream().cuda_stream


def ptr(t):
  if t is None:
    return None
  assert t.is_cuda, 'HIP ops need device tensors'
  return t.data_ptr()


def _chk(t):
  assert t.is_contiguous(), 'expected a contiguous tensor'
  return t


# ------------------------------------------------------------------------------------------------ GEMM / conv
def _scratch_key(device):
  """Scratch buffers are per (device, stream): the encoder branches and the weight-gradient lane run concurrently."""
  device = torch.device(device)
  return (str(device), torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0)


_SPLITK_WS = {}
_RETIRED = []  # scratch buffers replaced by larger ones: never freed, a captured hipGraph may still hold their raw pointers


def _replace_scratch(table, key, buf):
  old = table.get(key)
  if old is not None:
    _RETIRED.append(old)
  table[key] = buf
  return buf


_NO_CONV_SPLITK = _os.environ.get('TFPP_DEBUG_NO_CONV_SPLITK', '0') == '1'
SPLITK_WS_FLOATS = 16 << 20


def splitk_workspace(device):
  """fp32 slices of split-K launches (few output tiles x long reduction); one buffer per device, kernels on a stream run
  in order.  TFPP_SPLITK=0 disables split-K."""
  key = _scratch_key(device)
  buf = _SPLITK_WS.get(key)
  if buf is None:
    n = SPLITK_WS_FLOATS if _os.environ.get('TFPP_SPLITK', '1') != '0' else 4
    buf = torch.empty(n, device=device, dtype=torch.float32)
    _SPLITK_WS[key] = buf
  return buf


def conv_gemm(src, w, dst, *, B, Hs, Ws, Cs, Hd, Wd, Cd, R=1, S=1, stride=1, pad=0, G=1, ks_g=None, n_g=None, mode=0,
              act=ACT_NONE, scale=None, shift=None, res=None, alpha=1.0, dst_nchw=False, src_ld=None, dst_ld=None,
              res_ld=None, stats_ws=None, stats_acc=None, plan_only=False, bns=None, bns_query=False, stats_rows_query=False, stats_store=None,
              in_bn=None, in_relu=False, in_bn_query=False, relu_mask=None, relu_mask_query=False):
  """stats_ws (double[2*Cd]): also produce per-channel sum / sum of squares of the result (fused BatchNorm statistics).
  stats_acc (True, or zeroed fp32 accumulation rows from stats_rows_buffer()): leave the statistics in the rows for
  bn_finalize_partials and return (number of rows used, rows buffer).
  stats_store (fp32 [>= rows * 2 * Cd], the layer's own buffer): one row per M-tile, STORED (nothing to zero); returns (rows, buffer) -- the
  consumers of the normalised tensor finalize them in their prologues (tfpp_bn_rows).  stats_rows_query: only return that row count.
  in_bn (BnRows) / in_relu: the source is (raw, BatchNorm statistics), normalised while it is staged; in_bn_query: can this launch do that?"""
  p = ConvParams()
  p.src, p.w, p.dst = ptr(src), ptr(w), ptr(dst)
  p.scale, p.shift, p.res = ptr(scale), ptr(shift), ptr(res)
  p.B, p.Hs, p.Ws, p.Cs, p.Hd, p.Wd, p.Cd = B, Hs, Ws, Cs, Hd, Wd, Cd
  p.R, p.S, p.stride, p.pad, p.G = R, S, stride, pad, G
  p.ks_g = ks_g if ks_g is not None else Cs // G
  p.n_g = n_g if n_g is not None else Cd // G
  p.mode, p.act, p.dst_nchw, p.alpha = mode, act, int(dst_nchw), alpha
  p.src_ld = src_ld if src_ld is not None else Cs
  p.dst_ld = dst_ld if dst_ld is not None else Cd
  p.res_ld = res_ld if res_ld is not None else p.dst_ld
  p.dst_f32 = int(dst is not None and dst.dtype == torch.float32 and src.dtype != torch.float32)  # (dst None: geometry queries)
  ws = splitk_workspace(src.device)
  p.splitk_ws, p.splitk_ws_floats, p.splitk = ptr(ws), ws.numel(), 0
  if _NO_CONV_SPLITK:  # debugging aid: no K split for forward / data-gradient GEMMs while the weight-gradient slices keep their workspace
    p.splitk_ws, p.splitk_ws_floats = None, 0
  if stats_rows_query or in_bn_query:
    p.stats_partial = 16  # never dereferenced: plan as the launch with the fused BatchNorm statistics will be planned
    if in_bn_query:
      return bool(lib.raw('tfpp_conv_gemm_in_bn_ok')(ctypes.byref(p), dt(src)))
    return lib.raw('tfpp_conv_gemm_stats_rows')(ctypes.byref(p), dt(src))
  if in_bn is not None:
    p.in_bn, p.in_relu = in_bn, int(in_relu)
  if relu_mask_query:  # can the kernel that runs this data gradient apply a ReLU mask in its epilogue (tfpp_conv_params.relu_mask)?
    p.relu_mask_ld = Cd
    return bool(lib.raw('tfpp_conv_gemm_relu_mask_ok')(ctypes.byref(p), dt(src)))
  if relu_mask is not None:  # forward value of the tensor whose gradient this call completes: zero the result where it is <= 0
    p.relu_mask, p.relu_mask_ld = ptr(relu_mask), Cd
  if bns_query:  # (can the kernel that runs here emit the fused BatchNorm-backward statistics?, rows of bns_partial it would write)
    p.bns_ld = Cd
    p.bns_partial = 16  # never dereferenced: plan as the launch with the statistics will be planned (tile variant, no split-K)
    return bool(lib.raw('tfpp_conv_gemm_bns_ok')(ctypes.byref(p), dt(src))), lib.raw('tfpp_conv_gemm_stats_rows')(ctypes.byref(p), dt(src))
  if bns is not None:  # fused BatchNorm-backward statistics of the tensor whose gradient this call completes (tfpp.h)
    p.bns_y, p.bns_x, p.bns_mean, p.bns_invstd = ptr(bns['y']), ptr(bns['x']), ptr(bns['mean']), ptr(bns['invstd'])
    p.bns_partial, p.bns_ld, p.bns_relu = ptr(bns['partial']), Cd, int(bns['relu'])
  if plan_only:  # (kernel variant, K slices) the dispatcher would use -- tests / bench bookkeeping
    if stats_acc is not None or stats_ws is not None:
      p.stats_partial = 16  # never dereferenced: plan as the launch with the fused BatchNorm statistics will be planned
    return lib.raw('tfpp_conv_gemm_variant')(ctypes.byref(p), dt(src)), lib.raw('tfpp_conv_gemm_splits')(ctypes.byref(p), dt(src))
  scratch = None
  if stats_store is not None:
    p.stats_partial = 16
    nblk = lib.raw('tfpp_conv_gemm_stats_rows')(ctypes.byref(p), dt(src))
    assert stats_store.numel() >= nblk * 2 * Cd and stats_store.dtype == torch.float32
    p.stats_partial, p.stats_rows, p.stats_store = ptr(stats_store), nblk, 1
  elif stats_acc is not None:
    nblk = lib.raw('tfpp_conv_gemm_stats_rows')(ctypes.byref(p), dt(src))
    if STATS_ROWS_CAP > 0:
      nblk = min(STATS_ROWS_CAP, nblk)
    if stats_acc is True:
      stats_acc = stats_rows_buffer(Cd, src.device, nblk)
    assert stats_acc.numel() >= nblk * 2 * Cd
    p.stats_partial, p.stats_rows = ptr(stats_acc), nblk
  elif stats_ws is not None:
    nblk = min(64, lib.raw('tfpp_conv_gemm_mtiles')(ctypes.byref(p)))
    scratch = bn_scratch(Cd, src.device)
    lib.tfpp_zero(ptr(scratch), nblk * 2 * Cd * 4, stream())
    p.stats_partial, p.stats_rows = ptr(scratch), nblk
  if lib.profiler is not None:
    var = lib.raw('tfpp_conv_gemm_variant')(ctypes.byref(p), dt(src))
    if var >= 300:
      tile = f'halo8x32x{(var - 300) * 16}'
    elif var >= 200:
      tile = ('glds128x128', 'glds64x128', 'glds256x128')[var - 200]
    else:
      tile = ('128x32', '128x64', '64x64', '128x128', '128x96')[var]
    fam = f'conv_gemm<{"f32" if src.dtype == torch.float32 else "bf16"},{tile}>'
    if PROFILE_SHAPES:
      fam += f' m{mode} M={B * Hd * Wd} N={p.n_g} K={R * S * p.ks_g} G={G} k{R}s{stride}'
    esz = src.element_size()
    nbytes = (B * Hs * Ws * G * p.ks_g + G * p.n_g * R * S * p.ks_g + (B * Hd * Wd * G * p.n_g if res is not None else 0)) * esz + \
        B * Hd * Wd * G * p.n_g * dst.element_size()  # source, weights (, residual) read once; destination written once
    # the fusion-transformer linears (transfuser.py:352-359,383-402: forward and data gradient) are the only bf16 1x1 layers on token matrices
    grp = None
    if src.dtype == torch.bfloat16 and Hd == 1 and Wd == 1 and R == 1 and S == 1 and G == 1:
      grp = ('fusion_linears', 'fusion_linears_c1512') if min(p.n_g, p.ks_g) >= 1512 else ('fusion_linears',)  # (all four scales, the stage-4 transformer)
    lib.profiler.tag(fam, 2.0 * B * Hd * Wd * G * p.n_g * (R * S * p.ks_g), nbytes, group=grp)
  lib.tfpp_conv_gemm(ctypes.byref(p), dt(src), stream())
  if stats_store is not None:
    return nblk, stats_store
  if stats_acc is not None:
    return nblk, stats_acc
  if stats_ws is not None:
    lib.tfpp_bn_reduce_final(ptr(scratch), ptr(stats_ws), nblk, 2 * Cd, stream())
  return dst


def _wgrad_params(dy, x, dw, *, B, Hs, Ws, Cs, Hd, Wd, Cd, R=1, S=1, stride=1, pad=0, G=1, ks_g=None, n_g=None, c_real=None,
                  row_map=None, col_map=None, x_ld=None, dy_ld=None, dw_ld=None, splits=0, x_scale=None, x_shift=None, x_relu=False):
  p = WgradParams()
  p.x_scale, p.x_shift, p.x_relu = ptr(x_scale), ptr(x_shift), int(x_relu)
  p.dy, p.x, p.dw = ptr(dy), ptr(x), ptr(dw)
  p.row_map, p.col_map = ptr(row_map), ptr(col_map)
  p.B, p.Hs, p.Ws, p.Cs, p.Hd, p.Wd, p.Cd = B, Hs, Ws, Cs, Hd, Wd, Cd
  p.R, p.S, p.stride, p.pad, p.G = R, S, stride, pad, G
  p.ks_g = ks_g if ks_g is not None else Cs // G
  p.n_g = n_g if n_g is not None else Cd // G
  p.c_real = c_real if c_real is not None else p.ks_g
  p.splits = splits
  p.x_ld = x_ld if x_ld is not None else Cs
  p.dy_ld = dy_ld if dy_ld is not None else Cd
  p.dw_ld = dw_ld if dw_ld is not None else p.c_real * R * S
  assert dw.dtype == torch.float32
  ws = splitk_workspace(dy.device)
  p.ws, p.ws_floats = ptr(ws), ws.numel()
  return p


def conv_wgrad_x_bn_ok(dy, x, dw, **kw):
  """Can the kernel that runs this weight gradient normalise x while loading it (x_scale / x_shift / x_relu)?"""
  p = _wgrad_params(dy, x, dw, **kw)
  return bool(lib.raw('tfpp_conv_wgrad_x_bn_ok')(ctypes.byref(p), dt(dy)))


def conv_wgrad_plan(dy, x, dw, **kw):
  """(kernel variant, pixel slices, second-stage sum?) the dispatcher uses for this weight gradient -- tests / bench bookkeeping."""
  p = _wgrad_params(dy, x, dw, **kw)
  plan = (ctypes.c_int * 3)()
  lib.raw('tfpp_conv_wgrad_stage')(ctypes.byref(p), dt(dy), -1, plan, stream())
  return plan[0], plan[1], plan[2]


WGRAD_BATCH = None  # list of (params, tensors kept alive) while a batch of the weight-gradient lane is being collected
WGRAD_GROUP = _os.environ.get('TFPP_WGRAD_GROUP', '1') != '0'


def wgrad_batch_begin():
  """From here to wgrad_batch_end() the bf16 conv_wgrad calls of the CURRENT stream are collected instead of launched (the closures of one
  flush of engine.SideLane: independent layers), then handed to tfpp_conv_wgrad_batch in one call -- grouped grids instead of ~200 launches."""
  global WGRAD_BATCH
  if WGRAD_GROUP and WGRAD_BATCH is None:
    WGRAD_BATCH = []
    return True
  return False


def wgrad_batch_end():
  global WGRAD_BATCH
  batch, WGRAD_BATCH = WGRAD_BATCH, None
  if not batch:
    return
  def launch(items, family=None):
    arr = (WgradParams * len(items))()
    flops = nbytes = 0.0
    for i, p in enumerate(items):
      ctypes.memmove(ctypes.byref(arr, i * ctypes.sizeof(WgradParams)), ctypes.byref(p), ctypes.sizeof(WgradParams))
      kk = p.R * p.S * p.ks_g
      flops += 2.0 * p.B * p.Hd * p.Wd * p.G * p.n_g * kk
      # algorithmic bytes: dY and X read once, the gradient read-modify-written once in fp32
      nbytes += (p.B * p.Hd * p.Wd * p.G * p.n_g + p.B * p.Hs * p.Ws * p.G * p.ks_g) * 2 + 2 * p.G * p.n_g * kk * 4
    if family is not None:
      lib.profiler.tag(family, flops, nbytes)
    lib.tfpp_conv_wgrad_batch(arr, len(items), BF16, stream())

  items = [p for p, _ in batch]
  if lib.profiler is None:
    launch(items)
    return
  # profiling steps (bench.py roofline): the members of the batch are launched -- and so timed by the events around the library call -- per
  # kernel: the layers of the grouped 128 x 128 grid, those of the grouped 64 x 64 grid, everything else.  Same kernels, same results (the
  # groups never mix tile classes); the pixel splits of a group are chosen for the group it is launched with, as always.
  members = {'group256': [], 'group128': [], 'group64': [], 'single': []}
  plan = (ctypes.c_int * 3)()
  wide_on = _os.environ.get('TFPP_WGRAD_GROUP_256', '1') != '0'
  for p in items:
    lib.raw('tfpp_conv_wgrad_stage')(ctypes.byref(p), BF16, -1, plan, stream())
    ok = lib.raw('tfpp_conv_wgrad_group_ok')(ctypes.byref(p), BF16) if plan[0] in (2, 4) else 0
    wide = wide_on and ok and plan[0] == 4 and p.n_g >= 1024 and p.ks_g >= 1024  # (the routing of tfpp_conv_wgrad_batch: 256 x 256 tiles)
    members['group256' if wide else ('group128' if (ok and plan[0] == 4) else ('group64' if (ok and plan[0] == 2) else 'single'))].append(p)
  for k, its in members.items():
    if k == 'single':  # everything the grouped grids do not cover (3x3, narrow, strided layers): one profiled call per kernel, as outside a batch
      for p in its:
        _profiled_wgrad(p, BF16, 2)
    elif its:
      launch(its, f'conv_wgrad<bf16,{k}>')


_DBG_SKIP = set(_os.environ.get('TFPP_DEBUG_SKIP_OPS', '').split(','))  # timing experiments only (results are wrong): which side-lane work costs what?


def conv_wgrad(dy, x, dw, **kw):
  p = _wgrad_params(dy, x, dw, **kw)
  if _DBG_SKIP and (('wgrad_f32' in _DBG_SKIP and dy.dtype == torch.float32) or ('wgrad_narrow' in _DBG_SKIP and dy.dtype != torch.float32 and (p.n_g <= 32 or p.R * p.S * p.ks_g <= 32) and p.R == 1) or ('wgrad_3x3' in _DBG_SKIP and p.R == 3) or ('wgrad_all' in _DBG_SKIP)):
    return dw
  if WGRAD_BATCH is not None and dy.dtype == torch.bfloat16:
    WGRAD_BATCH.append((p, (dy, x, dw, kw.get('row_map'), kw.get('col_map'), kw.get('x_scale'), kw.get('x_shift'))))
    return dw
  if lib.profiler is not None:  # one profiled call per kernel: first stage and slice sum are timed separately
    _profiled_wgrad(p, dt(dy), dy.element_size())
    return dw
  lib.tfpp_conv_wgrad(ctypes.byref(p), dt(dy), stream())
  return dw


def _profiled_wgrad(p, dtc, esz):
  """One weight gradient under the bench.py profiler: first stage and slice sum as separately timed, separately tagged library calls."""
  B, Hd, Wd, Hs, Ws, G, R, S, stride = p.B, p.Hd, p.Wd, p.Hs, p.Ws, p.G, p.R, p.S, p.stride
  plan = (ctypes.c_int * 3)()
  lib.raw('tfpp_conv_wgrad_stage')(ctypes.byref(p), dtc, -1, plan, stream())  # plan only: no launch, not timed
  kind = ('lds32x32', 'lds64x64', 'glds64x64', 'halo3x3', 'glds128x128')[plan[0]]
  fam = f'conv_wgrad<{"f32" if dtc == F32 else "bf16"},{kind}>'
  if PROFILE_SHAPES:
    fam += f' P={B * Hd * Wd} N={p.n_g} K={R * S * p.ks_g} G={G} k{R}s{stride}'
  kk = R * S * p.ks_g
  # algorithmic bytes: dY and X read once, the fp32 slices (or the gradient itself, read-modify-write) written once
  nbytes = (B * Hd * Wd * G * p.n_g + B * Hs * Ws * G * p.ks_g) * esz + (plan[1] if plan[2] else 2) * G * p.n_g * kk * 4
  lib.profiler.tag(fam, 2.0 * B * Hd * Wd * G * p.n_g * kk, nbytes)
  lib.tfpp_conv_wgrad_stage(ctypes.byref(p), dtc, 1, None, stream())
  if plan[2]:
    lib.profiler.tag('wgrad_slice_sum', 0.0)
    lib.tfpp_conv_wgrad_stage(ctypes.byref(p), dtc, 2, None, stream())


def bgemm(A, B, C, *, M, N, K, lda, ldb, ldc, batch0=1, batch1=1, a_bs=(0, 0), b_bs=(0, 0), c_bs=(0, 0), a_km=False,
          b_km=False, alpha=1.0, beta=0.0, bias=None, act=ACT_NONE):
  p = BgemmParams()
  p.A, p.B, p.C, p.bias = ptr(A), ptr(B), ptr(C), ptr(bias)
  p.M, p.N, p.K, p.lda, p.ldb, p.ldc = M, N, K, lda, ldb, ldc
  p.a_bs0, p.a_bs1, p.b_bs0, p.b_bs1, p.c_bs0, p.c_bs1 = a_bs[0], a_bs[1], b_bs[0], b_bs[1], c_bs[0], c_bs[1]
  p.batch0, p.batch1, p.a_km, p.b_km, p.act = batch0, batch1, int(a_km), int(b_km), act
  p.c_f32 = int(C.dtype == torch.float32 and A.dtype != torch.float32)
  p.alpha, p.beta = alpha, beta
  if lib.profiler is not None:
    lib.profiler.tag(f'bgemm<{"f32" if A.dtype == torch.float32 else "bf16"}>', 2.0 * batch0 * batch1 * M * N * K)
  lib.tfpp_bgemm(ctypes.byref(p), dt(A), stream())
  return C


# ------------------------------------------------------------------------------------------------ fused attention
def _attn_params(q, k, v, o, lse, *, B, nh, T, d, ld_q, ld_kv, ld_o, scale, p_drop=0.0, seed=0, d_o=None, dq=None, dk=None, dv=None,
                 delta=None, debug_p=None):
  p = AttnParams()
  p.q, p.k, p.v, p.o, p.lse = ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse)
  p.d_o, p.dq, p.dk, p.dv, p.delta, p.debug_p = ptr(d_o), ptr(dq), ptr(dk), ptr(dv), ptr(delta), ptr(debug_p)
  p.B, p.nh, p.T, p.d, p.ld_q, p.ld_kv, p.ld_o = B, nh, T, d, ld_q, ld_kv, ld_o
  p.scale, p.p_drop, p.seed, p.seed_offset = scale, p_drop, seed, ptr(SEED_OFFSET)
  return p


def attn_supported(q, **kw):
  """True when the fused attention kernels (csrc/attention_kernels.hip) handle this problem: bf16, T a multiple of 64 up to 320,
  storage head dim a multiple of 8 up to 384.  Everything else (the fp32 11 x 65 planning decoder) keeps bgemm + softmax."""
  if not q.is_cuda:
    return False
  p = _attn_params(q, q, q, q, None, **kw)
  return bool(lib.raw('tfpp_attn_supported')(ctypes.byref(p), dt(q)))


def attn_fwd(q, k, v, o, lse, debug_p=None, **kw):
  p = _attn_params(q, k, v, o, lse, debug_p=debug_p, **kw)
  if lib.profiler is not None:
    lib.profiler.tag('attn_fwd<bf16,fused>', 4.0 * kw['B'] * kw['nh'] * kw['T'] * kw['T'] * kw['d'])
  lib.tfpp_attn_fwd(ctypes.byref(p), dt(q), stream())
  return o


def attn_bwd(q, k, v, o, lse, d_o, dq, dk, dv, delta, **kw):
  p = _attn_params(q, k, v, o, lse, d_o=d_o, dq=dq, dk=dk, dv=dv, delta=delta, **kw)
  if lib.profiler is not None:
    lib.profiler.tag('attn_bwd<bf16,fused>', 14.0 * kw['B'] * kw['nh'] * kw['T'] * kw['T'] * kw['d'])
  lib.tfpp_attn_bwd(ctypes.byref(p), dt(q), stream())


# ------------------------------------------------------------------------------------------------ packing
class PackPlan:
  """Records every weight-packing request once (persistent destinations) and replays them all as ONE launch."""

  def __init__(self):
    self.descs, self.keep, self.table, self.total_blocks = [], [], None, 0
    self.early = True     # tag of the descriptors added from now on: needed by the first layers of forward (stems, stage 1) or later
    self.tags = []
    self.tables = None    # [(device table, n descriptors, workgroups)] for the early / the late descriptors

  def add(self, kind, src, dst, total, a, row_map=None, col_map=None, in_ld=0, out_ld=0):
    self.tags.append(bool(self.early))
    d = PackDesc()
    d.src, d.dst, d.row_map, d.col_map = ptr(src), ptr(dst), ptr(row_map), ptr(col_map)
    d.total, d.in_ld, d.out_ld, d.kind, d.dtype = total, in_ld, out_ld, kind, dt(dst)
    for i, v in enumerate(a):
      d.a[i] = int(v)
    self.descs.append(d)
    self.keep += [src, dst, row_map, col_map]

  def finalize(self, device):
    plan = lib.raw('tfpp_pack_desc_plan')
    self.tables = []
    for want in (True, False):
      part = [d for d, t in zip(self.descs, self.tags) if t == want]
      blk = 0
      for d in part:
        d.blk_start = blk
        blk += plan(ctypes.byref(d))
      raw = b''.join(bytes(d) for d in part)
      self.tables.append((torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device) if part else None, len(part), blk))
    self.total_blocks = sum(t[2] for t in self.tables)

  def launch(self, part=None):
    """part None: every descriptor (two launches on the current stream); 0: the early table (stems, stage 1); 1: the rest."""
    for i, (table, n, blocks) in enumerate(self.tables or []):
      if n and (part is None or part == i):
        lib.tfpp_pack_multi(ptr(table), n, blocks, stream())


PACK_PLAN = None  # when set, pack_conv_weight / pack2d record into it instead of launching


def pack_conv_weight(w, dtype, G=1, ks_pad=None, n_pad=None, transpose=False):
  """OIHW fp32 parameter -> kernel image (see tfpp.h)."""
  cout, cin_g, r, s = w.shape
  n_g = cout // G
  ks_pad = ks_pad or cin_g
  n_pad = n_pad or n_g
  if transpose:
    out = torch.empty((G, cin_g, r * s * n_pad), device=w.device, dtype=dtype)
  else:
    out = torch.empty((G, n_pad, r * s * ks_pad), device=w.device, dtype=dtype)
  if PACK_PLAN is not None:
    PACK_PLAN.add(1 if transpose else 0, _chk(w), out, out.numel(), (cout, cin_g, r, s, G, ks_pad, n_pad))
    return out
  lib.tfpp_pack_conv_weight(ptr(_chk(w)), ptr(out), cout, cin_g, r, s, G, ks_pad, n_pad, int(transpose), dt(out), stream())
  return out


def pack2d(src, out, rows_out, cols_out, in_ld, out_ld, row_map=None, col_map=None, transpose_in=False, out_offset=0):
  """out[r][c] = src[rmap(r)][cmap(c)] (or transposed source); ``out_offset`` in elements."""
  o = out.view(-1)[out_offset:]
  if PACK_PLAN is not None:
    PACK_PLAN.add(2, src, o, rows_out * cols_out, (rows_out, cols_out, int(transpose_in)), row_map, col_map, in_ld, out_ld)
    return out
  lib.tfpp_pack2d(ptr(src), ptr(o), ptr(row_map), ptr(col_map), rows_out, cols_out, in_ld, out_ld, int(transpose_in), dt(out),
                  stream())
  return out


def cast(x, dtype):
  if x.dtype == dtype:
    return x
  out = torch.empty(x.shape, device=x.device, dtype=dtype)
  lib.tfpp_cast(ptr(_chk(x)), ptr(out), x.numel(), dt(x), dt(out), stream())
  return out


# ------------------------------------------------------------------------------------------------ layout
def nchw_to_nhwc_affine(x, dtype, cpad, mul=None, add=None):
  b, c, h, w = x.shape
  out = torch.empty((b, h, w, cpad), device=x.device, dtype=dtype)
  lib.tfpp_nchw_to_nhwc_affine(ptr(_chk(x)), ptr(out), ptr(mul), ptr(add), b, c, h, w, cpad, dt(out), stream())
  return out


def u8_to_nhwc_affine(x, dtype, cpad, mul=None, add=None, hwc=None, swap=False):
  """uint8 camera frames -> normalised NHWC (tfpp_u8_to_nhwc_affine); x: [B,H,W,C] (hwc) or [B,C,H,W]."""
  if hwc is None:
    hwc = x.shape[-1] <= 4
  b, (c, h, w) = x.shape[0], ((x.shape[3], x.shape[1], x.shape[2]) if hwc else x.shape[1:])
  out = torch.empty((b, h, w, cpad), device=x.device, dtype=dtype)
  lib.tfpp_u8_to_nhwc_affine(ptr(_chk(x)), ptr(out), ptr(mul), ptr(add), b, c, h, w, cpad, int(hwc), int(swap), dt(out), stream())
  return out


def nhwc_to_nchw(x, c_real, act=ACT_NONE):
  b, h, w, ld = x.shape
  out = torch.empty((b, c_real, h, w), device=x.device, dtype=torch.float32)
  lib.tfpp_nhwc_to_nchw(ptr(_chk(x)), ptr(out), b, c_real, h, w, ld, act, dt(x), stream())
  return out


def nchw_to_nhwc_pad(g, dtype, ld):
  b, c, h, w = g.shape
  out = torch.empty((b, h, w, ld), device=g.device, dtype=dtype)
  lib.tfpp_nchw_to_nhwc_pad(ptr(_chk(g)), ptr(out), b, c, h, w, ld, dt(out), stream())
  return out


# ------------------------------------------------------------------------------------------------ batch norm
_BN_SCRATCH = {}


def bn_scratch(c, device, min_floats=0):
  """Shared scratch for the BatchNorm reductions (kernels on one stream run in order, so one buffer serves every layer)."""
  need = max(lib.raw('tfpp_bn_scratch_floats')(c), min_floats)
  key = _scratch_key(device)
  buf = _BN_SCRATCH.get(key)
  if buf is None or buf.numel() < need:
    buf = _replace_scratch(_BN_SCRATCH, key, torch.empty(max(need, lib.raw('tfpp_bn_scratch_floats')(1512)), device=device, dtype=torch.float32))
  return buf


_STATS_ROWS = {}
_REDUCE_SCRATCH = {}
_GRIDSUM_SCRATCH = {}


def gridsum_scratch(device):
  """Ticket counters + partials of the fixed-order grid sums (include/tfpp.h tfpp_gridsum_scratch_floats): zero when handed out -- the kernels
  leave the counters at zero -- and one buffer per stream, like the other scratch tables."""
  key = _scratch_key(device)
  buf = _GRIDSUM_SCRATCH.get(key)
  if buf is None:
    if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
      raise RuntimeError('the grid-sum scratch must be allocated before hipGraph capture: run one eager warm-up step first')
    buf = _replace_scratch(_GRIDSUM_SCRATCH, key, zero_(torch.empty(lib.raw('tfpp_gridsum_scratch_floats')(), device=device, dtype=torch.float32)))
  return buf


STATS_ROWS_CAP = int(_os.environ.get('TFPP_BN_STATS_ROWS', '0'))  # 0: one row per M-tile (deterministic); n > 0: fold onto n rows


def stats_rows_buffer(c, device, rows=64):
  """fp32 accumulation rows [rows][2*c] for the BatchNorm statistics fused into the conv epilogue.  Zero when handed out
  and zero again after bn_finalize_partials(clear=True) consumed it, so no memset launch sits between layers.  With one
  row per M-tile (the default) every (row, channel) cell receives exactly one addend: the statistics -- and with them the
  whole bf16 training step -- are bit-reproducible run to run; folding M-tiles onto fewer rows adds them in atomic order."""
  key = _scratch_key(device)
  buf = _STATS_ROWS.get(key)
  need = rows * 2 * c
  if buf is None or buf.numel() < need:
    if torch.cuda.is_current_stream_capturing():
      # a buffer born inside a capture would be zeroed by that graph only, and a grown one would strand the pointers an earlier
      # capture holds: run one eager step of the same shapes before capturing (GraphedTrainStep / GraphedForward warm-ups do)
      raise RuntimeError('BatchNorm statistics rows must be allocated before hipGraph capture: run one eager warm-up step first')
    buf = _replace_scratch(_STATS_ROWS, key, zero_(torch.empty(max(need, 64 * 2 * 1512), device=device, dtype=torch.float32)))
  return buf


def clear_stats_rows(device):
  """Zero the accumulation rows of every stream of ``device`` (start of a training step, on the caller's stream: the lanes
  fork from it afterwards).  The rows are self-clearing in normal operation; this makes a step independent of an exception or
  an aborted capture having left addends behind."""
  dev = str(torch.device(device))
  for (d, _), buf in _STATS_ROWS.items():
    if d == dev or d.split(':')[0] == dev:
      zero_(buf)


def clone_scratch_for_current_stream(device):
  """Give the CURRENT stream its own set of scratch buffers (split-K / weight-gradient slices, BatchNorm partials and statistics
  rows, column-reduction partials), sized like the largest set any stream of ``device`` already owns.  graph.py calls this on the
  capture stream before a capture: the step has run eagerly on another stream by then, so every buffer the captured launches need
  exists (and the statistics rows are zeroed) before the capture begins -- nothing is allocated, zero-filled or grown inside it."""
  dev = str(torch.device(device))
  key = _scratch_key(device)
  for table, zeroed in ((_SPLITK_WS, False), (_BN_SCRATCH, False), (_STATS_ROWS, True), (_REDUCE_SCRATCH, False), (_GRIDSUM_SCRATCH, True)):
    sizes = [buf.numel() for (d, _), buf in table.items() if d == dev]
    if not sizes:
      continue
    cur = table.get(key)
    if cur is None or cur.numel() < max(sizes):
      buf = torch.empty(max(sizes), device=device, dtype=torch.float32)
      _replace_scratch(table, key, zero_(buf) if zeroed else buf)  # the replaced buffer stays alive: earlier captures point into it


def reduce_scratch(b, c, device):
  """Stage-1 partials of the two-stage column reductions (mean_hw / se_dgate / colsum)."""
  need = lib.raw('tfpp_reduce_scratch_floats')(b, c)
  key = _scratch_key(device)
  buf = _REDUCE_SCRATCH.get(key)
  if buf is None or buf.numel() < need:
    buf = _replace_scratch(_REDUCE_SCRATCH, key, torch.empty(max(need, lib.raw('tfpp_reduce_scratch_floats')(12, 1512)), device=device, dtype=torch.float32))
  return buf


def bn_finalize_partials(acc, nrows, gamma, beta, rm, rv, nbt, scale, shift, save_mean, save_invstd, rows, momentum=0.1, eps=1e-5,
                         clear=True):
  c = scale.numel()
  lib.tfpp_bn_finalize_partials(ptr(acc), nrows, int(clear), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(scale), ptr(shift),
                                ptr(save_mean), ptr(save_invstd), rows, c, momentum, eps, stream())


def bn_stats(x, ws):
  c = x.shape[-1]
  lib.tfpp_bn_stats(ptr(_chk(x)), ptr(bn_scratch(c, x.device)), ptr(ws), x.numel() // c, c, dt(x), stream())


def bn_finalize(ws, gamma, beta, rm, rv, nbt, scale, shift, save_mean, save_invstd, rows, momentum=0.1, eps=1e-5):
  c = scale.numel()
  lib.tfpp_bn_finalize(ptr(ws), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(scale), ptr(shift), ptr(save_mean),
                       ptr(save_invstd), rows, c, momentum, eps, stream())


def bn_fold(gamma, beta, rm, rv, scale, shift, eps=1e-5):
  lib.tfpp_bn_fold(ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(scale), ptr(shift), scale.numel(), eps, stream())


def affine_act(x, y=None, scale=None, shift=None, res=None, gate=None, rows_per_batch=1, act=ACT_NONE):
  c = x.shape[-1]
  if y is None:
    y = torch.empty_like(x)
  lib.tfpp_affine_act(ptr(_chk(x)), ptr(scale), ptr(shift), ptr(res), ptr(gate), ptr(y), x.numel() // c, c, rows_per_batch, act,
                      dt(x), stream())
  return y


def bn_bwd(dy, y, x, gamma, save_mean, save_invstd, ws, dgamma, dbeta, relu_mask, want_dres=False):
  c = x.shape[-1]
  rows = x.numel() // c
  scratch = bn_scratch(c, x.device)
  lib.tfpp_bn_bwd_reduce(ptr(_chk(dy)), ptr(y), ptr(_chk(x)), ptr(save_mean), ptr(save_invstd), ptr(scratch), None, rows, c,
                         int(relu_mask), dt(x), stream())
  dx = torch.empty_like(x)
  dres = torch.empty_like(x) if want_dres else None
  lib.tfpp_bn_bwd_apply(ptr(dy), ptr(y), ptr(x), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(ws), ptr(scratch), ptr(dx),
                        ptr(dres), ptr(dgamma), ptr(dbeta), rows, c, int(relu_mask), dt(x), stream())
  return dx, dres


def bn_bwd_rows(dy, y, x, gamma, save_mean, save_invstd, partial, nrows, dgamma, dbeta, relu_mask, want_dres=False):
  """bn_bwd with the stage-1 sums already produced by the kernel that wrote dy (conv_gemm(bns=...), se_bwd_apply_bns)."""
  c = x.shape[-1]
  rows = x.numel() // c
  coef = torch.empty(3 * c, device=x.device, dtype=torch.float32)
  dx = torch.empty_like(x)
  dres = torch.empty_like(x) if want_dres else None
  lib.tfpp_bn_bwd_apply_rows(ptr(_chk(dy)), ptr(y), ptr(_chk(x)), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(partial), nrows, ptr(coef),
                             ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), rows, c, int(relu_mask), dt(x), stream())
  return dx, dres


# ---- round 6: the statistics step lives in the prologue of the pass that needs it (csrc/bn_rows_kernels.hip, tfpp_bn_rows) ----------------
BN_ROWS_MAX = 256  # include/tfpp.h TFPP_BN_ROWS_MAX: most rows a consumer adds in its prologue


def bn_rows(C, scale, shift, partial=None, nrows=0, count=0, gamma=None, beta=None, rm=None, rv=None, nbt=None, save_mean=None, save_invstd=None,
            momentum=0.1, eps=1e-5):
  """tfpp_bn_rows: partial given -> the consumer finalizes from the rows (and writes scale / shift / saved and running statistics);
  partial None -> scale / shift are final and read.  The caller keeps the tensors alive."""
  b = BnRows()
  b.partial, b.nrows, b.C, b.count = ptr(partial), nrows, C, count
  b.gamma, b.beta, b.running_mean, b.running_var, b.num_batches_tracked = ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt)
  b.scale, b.shift, b.save_mean, b.save_invstd = ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd)
  b.momentum, b.eps = (0.1 if momentum is None else momentum), eps
  return b


def bn_apply_rows(x, bn, res=None, gate=None, rows_per_batch=1, relu_pre=False, relu_post=False, y=None):
  """y = [relu](([relu](x * scale + shift)) * gate + res) with the BatchNorm finalize in the prologue when bn carries rows."""
  c = x.shape[-1]
  if y is None:
    y = torch.empty_like(x)
  lib.tfpp_bn_apply_rows(ptr(_chk(x)), ctypes.byref(bn), ptr(res), ptr(gate), ptr(y), x.numel() // c, rows_per_batch, int(relu_pre), int(relu_post),
                         dt(x), stream())
  return y


MASK_NONE, MASK_Y, MASK_RAW = 0, 1, 2


def bn_bwd_reduce_rows(dy, y, x, scale, shift, save_mean, save_invstd, mask):
  """Stage-1 sums of the BatchNorm backward in the channel-block layout: returns (partial rows [nrows][2C], nrows)."""
  c = x.shape[-1]
  rows = x.numel() // c
  nrows = lib.raw('tfpp_bn_bwd_rows_count')(rows, c, dt(x))
  partial = torch.empty(nrows * 2 * c, device=x.device, dtype=torch.float32)
  lib.tfpp_bn_bwd_reduce_rows(ptr(_chk(dy)), ptr(y), ptr(_chk(x)), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), ptr(partial), rows, c, mask,
                              dt(x), stream())
  return partial, nrows


def bn_bwd_apply_rows2(dy, y, x, scale, shift, gamma, save_mean, save_invstd, partial, nrows, dgamma, dbeta, mask, want_dres=False):
  """dx (and dres = masked dy) from the stage-1 rows; the coefficient step runs in the kernel's prologue (nrows <= BN_ROWS_MAX)."""
  c = x.shape[-1]
  rows = x.numel() // c
  assert nrows <= BN_ROWS_MAX
  dx = torch.empty_like(x)
  dres = torch.empty_like(x) if want_dres else None
  lib.tfpp_bn_bwd_apply_rows2(ptr(_chk(dy)), ptr(y), ptr(_chk(x)), ptr(scale), ptr(shift), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(partial), nrows,
                              ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta), rows, c, mask, dt(x), stream())
  return dx, dres


def mean_hw_bn(x, bn):
  """mean over HW of relu(BN(x)) for x = raw convolution output [B,H,W,C]; bn carries rows -> finalize in the prologue."""
  b, h, w, c = x.shape
  assert b <= 64
  out = torch.empty((b, c), device=x.device, dtype=torch.float32)
  lib.tfpp_mean_hw_bn(ptr(_chk(x)), ctypes.byref(bn), ptr(out), ptr(reduce_scratch(b, c, x.device)), ptr(gridsum_scratch(x.device)), b, h * w, dt(x), stream())
  return out


def se_bwd_apply_bn(dy, gate, dpool, x, scale, shift, save_mean, save_invstd):
  """dx = dy * gate + dpool / HW and the BatchNorm-backward rows of the layer in front (mask recomputed from the raw tensor x);
  returns (dx, partial rows, nrows)."""
  b, h, w, c = dy.shape
  nrows = lib.raw('tfpp_se_bwd_apply_bn_rows')(b, h * w, c, dt(dy))
  partial = torch.empty(nrows * 2 * c, device=dy.device, dtype=torch.float32)
  dx = torch.empty_like(dy)
  lib.tfpp_se_bwd_apply_bn(ptr(_chk(dy)), ptr(gate), ptr(dpool), ptr(_chk(x)), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), ptr(dx), ptr(partial),
                           b, h * w, c, dt(dy), stream())
  return dx, partial, nrows


# ---- SyncBatchNorm (team_code/train.py:511-512, config.sync_batch_norm = 1): statistics over the batches of ALL ranks.  The per-channel sums
# (2C doubles per layer and pass) are all-reduced between the kernels; everything else is the per-rank path.  Collectives inside the pass:
# such a step runs eagerly (graph.py / dropin.py refuse to capture it).
def bn_sync_finalize(acc, nrows, gamma, beta, rm, rv, nbt, scale, shift, save_mean, save_invstd, rows, world, group, ws, momentum=0.1, eps=1e-5):
  """bn_finalize_partials with the sums of every rank: rows = this rank's B*H*W (equal on all ranks, as DistributedSampler makes them)."""
  import torch.distributed as dist
  c = scale.numel()
  lib.tfpp_bn_reduce_final(ptr(acc), ptr(ws), nrows, 2 * c, stream())
  lib.tfpp_zero(ptr(acc), nrows * 2 * c * 4, stream())  # (the rows start every layer from zero: bn_finalize_partials(clear=True) does this in the per-rank path)
  dist.all_reduce(ws, op=dist.ReduceOp.SUM, group=group)
  lib.tfpp_bn_finalize(ptr(ws), ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(scale), ptr(shift), ptr(save_mean), ptr(save_invstd), rows * world, c,
                       momentum, eps, stream())


def bn_bwd_sync(dy, y, x, gamma, save_mean, save_invstd, dgamma, dbeta, relu_mask, world, group, want_dres=False):
  """bn_bwd with sum g / sum g*xhat over ALL ranks in dx (torch SyncBatchNorm's backward) and this rank's own sums in dgamma / dbeta
  (DistributedDataParallel averages the parameter gradients afterwards, as for every other parameter)."""
  import torch.distributed as dist
  c = x.shape[-1]
  rows = x.numel() // c
  scratch = bn_scratch(c, x.device)
  ws = torch.empty(2 * c, device=x.device, dtype=torch.float64)
  lib.tfpp_bn_bwd_reduce(ptr(_chk(dy)), ptr(y), ptr(_chk(x)), ptr(save_mean), ptr(save_invstd), ptr(scratch), ptr(ws), rows, c, int(relu_mask), dt(x), stream())
  local = torch.empty(2 * c, device=x.device, dtype=torch.float32)
  lib.tfpp_f64_to_f32(ptr(ws), ptr(local), 2 * c, 1.0, stream())
  dist.all_reduce(ws, op=dist.ReduceOp.SUM, group=group)
  glob = torch.empty(2 * c, device=x.device, dtype=torch.float32)
  # the coefficients of dx depend on sum / count only: the sums over all ranks divided by `world`, with this rank's row count (which the
  # kernel also uses as the extent of the tensor it walks)
  lib.tfpp_f64_to_f32(ptr(ws), ptr(glob), 2 * c, 1.0 / world, stream())
  coef = torch.empty(3 * c, device=x.device, dtype=torch.float32)
  dx = torch.empty_like(x)
  dres = torch.empty_like(x) if want_dres else None
  lib.tfpp_bn_bwd_apply_rows(ptr(dy), ptr(y), ptr(x), ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(glob), 1, ptr(coef), ptr(dx), ptr(dres), None, None,
                             rows, c, int(relu_mask), dt(x), stream())
  if dbeta is not None:
    copy_rows(local, dbeta, 1, c, 0, 0, 0, 0, accumulate=True)
  if dgamma is not None:
    copy_rows(local[c:], dgamma, 1, c, 0, 0, 0, 0, accumulate=True)
  return dx, dres


def instance_norm_fwd(x, act=ACT_NONE, eps=1e-5):
  """nn.InstanceNorm2d (affine=False, no running statistics) on NHWC x, optionally fused with ReLU: BatchNorm arithmetic on each sample's
  rows (team_code/bev_encoder.py:120,262-267).  Returns (y, saved mean [B,C], saved invstd [B,C])."""
  B, H, W, C = x.shape
  dev = x.device
  ws = torch.empty((B, 2 * C), device=dev, dtype=torch.float64)
  scale, shift, mean, invstd = (torch.empty((B, C), device=dev, dtype=torch.float32) for _ in range(4))
  y = torch.empty_like(x)
  for b in range(B):
    bn_stats(x[b], ws[b])
    bn_finalize(ws[b], None, None, None, None, None, scale[b], shift[b], mean[b], invstd[b], H * W, eps=eps)
    affine_act(x[b], y[b], scale=scale[b], shift=shift[b], act=act)
  return y, mean, invstd


def instance_norm_bwd(dy, y, x, mean, invstd, relu):
  B, H, W, C = x.shape
  dy = dy.view(B, H, W, C)
  dx = torch.empty_like(x)
  scratch = bn_scratch(C, x.device)
  for b in range(B):
    lib.tfpp_bn_bwd_reduce(ptr(dy[b]), ptr(y[b]), ptr(x[b]), ptr(mean[b]), ptr(invstd[b]), ptr(scratch), None, H * W, C, int(relu), dt(x), stream())
    lib.tfpp_bn_bwd_apply(ptr(dy[b]), ptr(y[b]), ptr(x[b]), None, ptr(mean[b]), ptr(invstd[b]), None, ptr(scratch), ptr(dx[b]), None, None, None,
                          H * W, C, int(relu), dt(x), stream())
  return dx


def bn1d_scalar(x, rm, rv, nbt, training, momentum=0.1, eps=1e-5):
  y = torch.empty_like(x)
  lib.tfpp_bn1d_scalar(ptr(_chk(x)), ptr(y), ptr(rm), ptr(rv), ptr(nbt), x.numel(), int(training), momentum, eps, stream())
  return y


# ------------------------------------------------------------------------------------------------ squeeze-excite
HW_TICKET = _os.environ.get('TFPP_HW_TICKET', '1') != '0'  # 0: the two-launch reductions (A/B runs, comparison tests)


def mean_hw(x):
  b, h, w, c = x.shape
  out = torch.empty((b, c), device=x.device, dtype=torch.float32)
  if HW_TICKET and b <= 64:  # one launch: the last workgroup of a sample adds the partial sums (tfpp_mean_hw_ticket)
    lib.tfpp_mean_hw_ticket(ptr(_chk(x)), ptr(out), ptr(reduce_scratch(b, c, x.device)), ptr(gridsum_scratch(x.device)), b, h * w, c, dt(x), stream())
    return out
  lib.tfpp_mean_hw(ptr(_chk(x)), ptr(out), ptr(reduce_scratch(b, c, x.device)), b, h * w, c, dt(x), stream())
  return out


def se_gate_fwd(pool, w1, b1, w2, b2):
  b, c = pool.shape
  rd = w1.shape[0]
  hidden = torch.empty((b, rd), device=pool.device, dtype=torch.float32)
  gate = torch.empty((b, c), device=pool.device, dtype=torch.float32)
  lib.tfpp_se_gate_fwd(ptr(pool), ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(hidden), ptr(gate), b, c, rd, stream())
  return hidden, gate


def se_dgate(dy, x):
  b, h, w, c = x.shape
  out = torch.empty((b, c), device=x.device, dtype=torch.float32)
  if HW_TICKET and b <= 64:
    lib.tfpp_se_dgate_ticket(ptr(_chk(dy)), ptr(_chk(x)), ptr(out), ptr(reduce_scratch(b, c, x.device)), ptr(gridsum_scratch(x.device)), b, h * w, c, dt(x), stream())
    return out
  lib.tfpp_se_dgate(ptr(_chk(dy)), ptr(_chk(x)), ptr(out), ptr(reduce_scratch(b, c, x.device)), b, h * w, c, dt(x), stream())
  return out


def se_gate_bwd(dgate, gate, hidden, pool, w1, w2, dw1, db1, dw2, db2, premul=False):
  """premul: ``dgate`` is dgate * gate = sum_hw dy * (gated tensor) (tfpp_se_gate_bwd_premul)"""
  b, c = gate.shape
  rd = hidden.shape[1]
  dpool = torch.empty_like(gate)
  scratch = torch.empty_like(hidden)
  fn = lib.tfpp_se_gate_bwd_premul if premul else lib.tfpp_se_gate_bwd
  fn(ptr(dgate), ptr(gate), ptr(hidden), ptr(pool), ptr(w1), ptr(w2), ptr(scratch), ptr(dpool), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), b, c, rd, stream())
  return dpool


def se_bwd_apply(dy, gate, dpool):
  b, h, w, c = dy.shape
  dx = torch.empty_like(dy)
  lib.tfpp_se_bwd_apply(ptr(_chk(dy)), ptr(gate), ptr(dpool), ptr(dx), b, h * w, c, dt(dy), stream())
  return dx


def se_bwd_apply_bns(dy, gate, dpool, y, x, save_mean, save_invstd):
  """se_bwd_apply that also emits the BatchNorm-backward sums of the layer in front of the squeeze-excite block; returns
  (dx, partial rows, number of rows)."""
  b, h, w, c = dy.shape
  nrows = lib.raw('tfpp_se_bwd_apply_bns_rows')(b, h * w, c, dt(dy))
  partial = torch.empty(nrows * 2 * c, device=dy.device, dtype=torch.float32)
  dx = torch.empty_like(dy)
  lib.tfpp_se_bwd_apply_bns(ptr(_chk(dy)), ptr(gate), ptr(dpool), ptr(_chk(y)), ptr(_chk(x)), ptr(save_mean), ptr(save_invstd), ptr(dx), ptr(partial),
                            b, h * w, c, dt(dy), stream())
  return dx, partial, nrows


# ------------------------------------------------------------------------------------------------ pooling / resampling
def avgpool_fwd(x, ho, wo, out=None, y_ld=None):
  b, h, w, c = x.shape
  if out is None:
    out = torch.empty((b, ho, wo, c), device=x.device, dtype=x.dtype)
  lib.tfpp_avgpool_fwd(ptr(_chk(x)), ptr(out), b, h, w, c, ho, wo, y_ld or c, dt(x), stream())
  return out


def avgpool_bwd_add(dy, dx, ho, wo, dy_ld=None):
  b, h, w, c = dx.shape
  lib.tfpp_avgpool_bwd_add(ptr(dy), ptr(_chk(dx)), b, h, w, c, ho, wo, dy_ld or c, dt(dx), stream())
  return dx


def bilinear_fwd(x, ho, wo, base=None, mul=None, x_ld=None, nchw_f32=False, c_real=None, shape_in=None):
  """x: [B,Hi,Wi,C] (or flat with shape_in=(B,Hi,Wi,C) and pixel stride x_ld)."""
  b, hi, wi, c = shape_in if shape_in is not None else x.shape
  if nchw_f32:
    c_real = c_real or c
    y = torch.empty((b, c_real, ho, wo), device=x.device, dtype=torch.float32)
  else:
    y = torch.empty((b, ho, wo, c), device=x.device, dtype=x.dtype)
  lib.tfpp_bilinear_fwd(ptr(x), ptr(base), ptr(mul), ptr(y), b, hi, wi, ho, wo, c, x_ld or c, c, int(nchw_f32), c_real or c, dt(x),
                        stream())
  return y


def bilinear_bwd(dy, hi, wi, mul=None, dx=None, dx_ld=None):
  b, ho, wo, c = dy.shape
  if dx is None:
    dx = torch.empty((b, hi, wi, c), device=dy.device, dtype=dy.dtype)
  lib.tfpp_bilinear_bwd(ptr(_chk(dy)), ptr(mul), ptr(dx), b, hi, wi, ho, wo, c, c, dx_ld or c, dt(dy), stream())
  return dx


# ------------------------------------------------------------------------------------------------ token ops
SEED_OFFSET = None  # device uint64 counter added to every dropout seed (set by the trainer)


def set_seed_offset(t):
  global SEED_OFFSET
  SEED_OFFSET = t


def inc_u64(t):
  lib.tfpp_inc_u64(ptr(t), stream())


def layernorm_fwd(x, gamma, beta, eps=1e-5, save=True):
  c = x.shape[-1]
  rows = x.numel() // c
  y = torch.empty_like(x)
  mean = torch.empty(rows, device=x.device, dtype=torch.float32) if save else None
  rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if save else None
  lib.tfpp_layernorm_fwd(ptr(_chk(x)), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, c, eps, dt(x), stream())
  return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma, dbeta):
  c = x.shape[-1]
  dx = torch.empty_like(x)
  scratch = gridsum_scratch(x.device) if (dgamma is not None or dbeta is not None) else None
  lib.tfpp_layernorm_bwd(ptr(_chk(dy)), ptr(_chk(x)), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(scratch),
                         x.numel() // c, c, dt(x), stream())
  return dx


def layernorm_param_grad(dy, x, mean, rstd, dgamma, dbeta):
  if 'ln_param' in _DBG_SKIP:
    return
  c = x.shape[-1]
  lib.tfpp_layernorm_param_grad(ptr(_chk(dy)), ptr(_chk(x)), ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta), ptr(gridsum_scratch(x.device)),
                                x.numel() // c, c, dt(x), stream())


def add_layernorm_fwd(a, b, gamma, beta, eps=1e-5, p_drop=0.0, seed=0, save=True):
  """sum = a + dropout(b); y = LayerNorm(sum).  Returns (y, sum, mean, rstd)."""
  c = a.shape[-1]
  rows = a.numel() // c
  y, s = torch.empty_like(a), torch.empty_like(a)
  mean = torch.empty(rows, device=a.device, dtype=torch.float32) if save else None
  rstd = torch.empty(rows, device=a.device, dtype=torch.float32) if save else None
  lib.tfpp_add_layernorm_fwd(ptr(_chk(a)), ptr(_chk(b)), ptr(s), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, c, eps, p_drop, seed,
                             ptr(SEED_OFFSET), dt(a), stream())
  return y, s, mean, rstd


def add_layernorm_bwd(dy, s, gamma, mean, rstd, dgamma, dbeta, p_drop=0.0, seed=0):
  """Returns (d_sum, d_b): the gradients of a and b in LayerNorm(a + dropout(b))."""
  c = s.shape[-1]
  ds, db = torch.empty_like(s), torch.empty_like(s)
  scratch = gridsum_scratch(s.device) if (dgamma is not None or dbeta is not None) else None
  lib.tfpp_add_layernorm_bwd(ptr(_chk(dy)), ptr(_chk(s)), ptr(gamma), ptr(mean), ptr(rstd), ptr(ds), ptr(db), ptr(dgamma), ptr(dbeta), ptr(scratch),
                             s.numel() // c, c, p_drop, seed, ptr(SEED_OFFSET), dt(s), stream())
  return ds, db


def softmax_fwd(x, rows, cols, ld, alpha=1.0, p_drop=0.0, seed=0):
  """In place; returns (P, P_dropped) -- the same tensor when p_drop == 0."""
  pd = torch.empty_like(x) if p_drop > 0.0 else None
  lib.tfpp_softmax_fwd(ptr(x), ptr(pd), rows, cols, ld, alpha, p_drop, seed, ptr(SEED_OFFSET), dt(x), stream())
  return x, (pd if pd is not None else x)


def small_attn_supported(tq, tk, d, dtype):
  return dtype == torch.float32 and bool(lib.raw('tfpp_small_attn_supported')(tq, tk, d))


def small_attn_fwd(q, k, v, o, p_save, *, B, nh, tq, tk, d, ld_q, ld_kv, ld_o, scale, p_drop=0.0, seed=0):
  """Attention core of the fp32 planning decoder as one launch (csrc/head_kernels.hip)."""
  if lib.profiler is not None:
    lib.profiler.tag('small_attn_fwd<f32>', 4.0 * B * nh * tq * tk * d)
  lib.tfpp_small_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(p_save), B, nh, tq, tk, d, ld_q, ld_kv, ld_o, scale, p_drop, seed, ptr(SEED_OFFSET), stream())
  return o


def small_attn_bwd(q, k, v, p_save, d_o, dq, dk, dv, *, B, nh, tq, tk, d, ld_q, ld_kv, ld_o, scale, p_drop=0.0, seed=0):
  if lib.profiler is not None:
    lib.profiler.tag('small_attn_bwd<f32>', 10.0 * B * nh * tq * tk * d)
  lib.tfpp_small_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(p_save), ptr(_chk(d_o)), ptr(dq), ptr(dk), ptr(dv), B, nh, tq, tk, d, ld_q, ld_kv, ld_o, scale,
                          p_drop, seed, ptr(SEED_OFFSET), stream())


def softmax_bwd(p, dp, rows, cols, ld, alpha=1.0, p_drop=0.0, seed=0):
  lib.tfpp_softmax_bwd(ptr(p), ptr(dp), rows, cols, ld, alpha, p_drop, seed, ptr(SEED_OFFSET), dt(p), stream())
  return dp


def patchify3d(x, dtype):
  """x: fp32 (B, T, H, W) -> [B * T/2 * H/4 * W/4, 32] (video_swin_transformer.py:427-467)."""
  B, T, H, W = x.shape
  out = torch.empty((B * (T // 2) * (H // 4) * (W // 4), 32), device=x.device, dtype=dtype)
  lib.tfpp_patchify3d(ptr(_chk(x)), ptr(out), B, T, H, W, dt(out), stream())
  return out


def gather_rows(src, idx, rows, C, add=None, out=None, src_ld=None, dst_ld=None, dst_off=0):
  """out[r, dst_off : dst_off + C] = src[idx[r], :C] (zeros where idx[r] < 0) (+ add[r, :C]); idx: int32 device tensor."""
  if out is None:
    out = torch.empty((rows, C), device=src.device, dtype=src.dtype)
  dst_ld = dst_ld or C
  lib.tfpp_gather_rows(ptr(src), ptr(idx), ptr(add), ptr(out.view(-1)[dst_off:]), rows, C, src_ld or C, dst_ld, C, dt(src), stream())
  return out


def softmax_window_bias(s, table, rel_index, mask, windows, heads, n, alpha, ld=None):
  """In place on s [windows, heads, n, ld] (video_swin_transformer.py:146-163)."""
  lib.tfpp_softmax_window_bias(ptr(s), ptr(table), ptr(rel_index), ptr(mask), windows, heads, n, ld or n, mask.shape[0] if mask is not None else 1,
                               alpha, dt(s), stream())
  return s


def bev_lift_fwd(feat, coords, scale, D, W, Z):
  """feat [B, Hf, Wf, C] -> [B, W, D, C] (team_code/bev_encoder.py:180-201)."""
  B, Hf, Wf, C = feat.shape
  out = torch.empty((B, W, D, C), device=feat.device, dtype=feat.dtype)
  lib.tfpp_bev_lift_fwd(ptr(_chk(feat)), ptr(coords), ptr(scale), ptr(out), B, Hf, Wf, C, D, W, Z, dt(feat), stream())
  return out


def bev_lift_bwd(dout, coords, scale, Hf, Wf, D, W, Z):
  B, _, _, C = dout.shape
  dfeat = zeros((B, Hf, Wf, C), torch.float32, dout.device)
  lib.tfpp_bev_lift_bwd(ptr(_chk(dout)), ptr(coords), ptr(scale), ptr(dfeat), B, Hf, Wf, C, D, W, Z, dt(dout), stream())
  return dfeat


def window_bias_dense(table, rel_index, heads, n, ld_b):
  """relative_position_bias_table[relative_position_index] as dense fp32 [heads, n, ld_b] (columns >= n zero)."""
  dense = torch.empty((heads, n, ld_b), device=table.device, dtype=torch.float32)
  lib.tfpp_window_bias_dense(ptr(table), ptr(rel_index), ptr(dense), heads, n, ld_b, stream())
  return dense


def attn_window_fwd(q, k, v, o, bias, mask, p_out, *, B, nh, T, d, ld_q, ld_kv, ld_o, scale):
  """Fused window attention (bf16): o = softmax(scale q k^T + bias[h] + mask[b % n_mask]) v per (window, head); p_out (optional, bf16
  [B, nh, T, ld_p]) receives the probabilities for the backward."""
  p = AttnParams()
  p.q, p.k, p.v, p.o = ptr(q), ptr(k), ptr(v), ptr(o)
  p.B, p.nh, p.T, p.d, p.ld_q, p.ld_kv, p.ld_o, p.scale = B, nh, T, d, ld_q, ld_kv, ld_o, scale
  p.bias, p.mask, p.p_out = ptr(bias), ptr(mask), ptr(p_out)
  p.n_mask = mask.shape[0] if mask is not None else 1
  p.ld_b = bias.shape[-1]
  p.ld_p = p_out.shape[-1] if p_out is not None else 0
  if lib.profiler is not None:
    lib.profiler.tag('attn_window_fwd<bf16,fused>', 4.0 * B * nh * T * T * d)
  lib.tfpp_attn_window_fwd(ctypes.byref(p), dt(q), stream())
  return o


def drop_path(x, samples, p, seed):
  """timm DropPath: one keep / drop draw per sample (video_swin_transformer.py:216,276-281)."""
  y = torch.empty_like(x)
  lib.tfpp_drop_path(ptr(_chk(x)), ptr(y), samples, x.numel() // samples, p, seed, ptr(SEED_OFFSET), dt(x), stream())
  return y


_WINDOW_BIAS_INV = {}  # (data_ptr of the device index, n) -> (inv_ptr, inv_pairs, table entries) on the device


def window_bias_grad(ds, rel_index, dtable, windows, heads, n, ld, scale):
  """dtable[rel_index[i, j], h] += scale * sum_w ds[w, h, i, j] without atomics: a dense (heads, n, n) window sum, then per table entry the pairs
  of an inverse index (built once per index tensor on the host, outside any capture) in a fixed order."""
  key = (rel_index.data_ptr(), n, str(rel_index.device))
  inv = _WINDOW_BIAS_INV.get(key)
  ntab = dtable.shape[0]
  if inv is None:
    if ds.is_cuda and torch.cuda.is_current_stream_capturing():
      raise RuntimeError('the inverse relative-position index must be built before hipGraph capture: run one eager warm-up step first')
    flat = rel_index.reshape(-1).to('cpu', torch.int64)
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=ntab)
    inv_ptr = torch.zeros(ntab + 1, dtype=torch.int32)
    inv_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    inv = _WINDOW_BIAS_INV[key] = (inv_ptr.to(ds.device), order.to(torch.int32).to(ds.device), rel_index)  # (keeps the index tensor alive: the key is its address)
  dense = torch.empty(heads * n * n, device=ds.device, dtype=torch.float32)
  lib.tfpp_window_bias_grad(ptr(ds), ptr(inv[0]), ptr(inv[1]), ntab, ptr(dense), ptr(dtable), windows, heads, n, ld, scale, dt(ds), stream())


def add_dropout(a, b, p_drop=0.0, seed=0, out=None):
  if out is None:
    out = torch.empty_like(b)
  lib.tfpp_add_dropout(ptr(a), ptr(_chk(b)), ptr(out), b.numel(), p_drop, seed, ptr(SEED_OFFSET), dt(b), stream())
  return out


def add_bcast(x, bc, out=None):
  if out is None:
    out = torch.empty_like(x)
  lib.tfpp_add_bcast(ptr(_chk(x)), ptr(bc), ptr(out), x.numel(), bc.numel(), dt(x), stream())
  return out


def act_bwd(dy, y, act, out=None):
  if out is None:
    out = torch.empty_like(dy)
  lib.tfpp_act_bwd(ptr(_chk(dy)), ptr(_chk(y)), ptr(out), dy.numel(), act, dt(dy), stream())
  return out


def axpy(x, y, a=1.0):
  lib.tfpp_axpy(ptr(_chk(x)), ptr(_chk(y)), x.numel(), a, dt(x), stream())
  return y


def mul_pixmask(x, m, hw, out=None):
  if out is None:
    out = torch.empty_like(x)
  lib.tfpp_mul_pixmask(ptr(_chk(x)), ptr(m), ptr(out), x.numel(), x.shape[-1], hw, dt(x), stream())
  return out


def colsum(x, out, rows, c, ld=None):
  if 'colsum' in _DBG_SKIP:
    return out
  lib.tfpp_colsum(ptr(x), ptr(out), ptr(reduce_scratch(1, c, x.device)), rows, c, ld or c, dt(x), stream())
  return out


def copy_rows(src, dst, B, n, src_bs, src_off, dst_bs, dst_off, accumulate=False):
  lib.tfpp_copy_rows(ptr(src), ptr(dst), B, n, src_bs, src_off, dst_bs, dst_off, int(accumulate), dt(src), dt(dst), stream())
  return dst


def zero_(t):
  lib.tfpp_zero(ptr(t), t.numel() * t.element_size(), stream())
  return t


def zeros(shape, dtype=torch.float32, device='cuda'):
  return zero_(torch.empty(shape, device=device, dtype=dtype))


# debugging aid (tools/replay_bisect.py): one order-independent 64-bit hash per tape event, written into a device table by kernels that are part of
# the captured step, so two replays of ONE hipGraph can be compared event by event
NODE_HASH = {'on': _os.environ.get('TFPP_DEBUG_NODE_HASH', '0') == '1', 'buf': None, 'n': 0, 'labels': [], 'lo': 0, 'hi': 1 << 30, 'slots': 32768}
if _os.environ.get('TFPP_DEBUG_NODE_HASH_RANGE'):
  NODE_HASH['lo'], NODE_HASH['hi'] = (int(v) for v in _os.environ['TFPP_DEBUG_NODE_HASH_RANGE'].split(':'))


STAMPS = {'on': _os.environ.get('TFPP_DEBUG_STAMPS', '0') == '1', 'buf': None, 'labels': [], 'n': 0}


def stamp(label):
  """Time stamp of the current stream at this point of the launch sequence (TFPP_DEBUG_STAMPS=1; tools/lane_timeline.py reads the table)."""
  st = STAMPS
  if not st['on']:
    return
  if st['buf'] is None:
    st['buf'] = zeros(4096, torch.int64, 'cuda')
  i = st['n']
  if i >= 4096:
    return
  st['n'] += 1
  if len(st['labels']) <= i:
    st['labels'].append(label)
  else:
    st['labels'][i] = label
  lib.tfpp_stamp(st['buf'].data_ptr() + 8 * i, stream())


def node_hash_begin(device):
  st = NODE_HASH
  if st['buf'] is None:
    st['buf'] = torch.empty(st['slots'], device=device, dtype=torch.int64)
  zero_(st['buf'])
  st['n'] = 0
  st['labels'] = []


def node_hash(t, label):
  st = NODE_HASH
  if t is None or not t.is_cuda or not t.is_contiguous() or st['buf'] is None:
    return
  i = st['n']
  st['n'] += 1
  st['labels'].append(f'{label} {tuple(t.shape)} {str(t.dtype)[6:]}')
  nbytes = t.numel() * t.element_size()
  if i >= st['slots'] or not (st['lo'] <= i < st['hi']) or nbytes % 4:
    return
  lib.tfpp_hash_words(ptr(t), nbytes, st['buf'].data_ptr() + 8 * i, stream())


def sum_f32(x, out):
  lib.tfpp_sum_f32(ptr(_chk(x)), ptr(out), x.numel(), stream())
  return out


# ------------------------------------------------------------------------------------------------ GRU / losses / optimizer
def gru_fwd(gi, h0, w_hh, b_hh, w_dec, b_dec):
  b, t, h3 = gi.shape
  h = h3 // 3
  save = torch.empty((b, t, 4, h), device=gi.device, dtype=torch.float32)
  out = torch.empty((b, t, 2), device=gi.device, dtype=torch.float32)
  lib.tfpp_gru_fwd(ptr(_chk(gi)), ptr(_chk(h0)), ptr(w_hh), ptr(b_hh), ptr(w_dec), ptr(b_dec), ptr(save), ptr(out), b, t, h, stream())
  return out, save


def gru_bwd(dout, save, h0, w_hh, b_hh, w_dec, dw_hh, db_hh, dw_dec, db_dec, defer=False):
  """defer=True: the parameter gradients are left as per-sample partial images; returns (dgi, dh0, partial, reduce) and the caller runs
  ``reduce()`` on whatever stream it likes once ``partial`` is complete there."""
  b, t, _, h = save.shape
  dgi = torch.empty((b, t, 3 * h), device=save.device, dtype=torch.float32)
  dh0 = torch.empty((b, h), device=save.device, dtype=torch.float32)
  part = torch.empty(int(lib.raw('tfpp_gru_bwd_partial_floats')(b, h)), device=save.device, dtype=torch.float32)
  dst = (None,) * 4 if defer else (dw_hh, db_hh, dw_dec, db_dec)
  lib.tfpp_gru_bwd(ptr(_chk(dout)), ptr(save), ptr(h0), ptr(w_hh), ptr(b_hh), ptr(w_dec), ptr(dgi), ptr(dh0), ptr(part), *(ptr(x) for x in dst),
                   b, t, h, stream())
  if not defer:
    return dgi, dh0
  return dgi, dh0, part, lambda: lib.tfpp_gru_bwd_reduce(ptr(part), ptr(dw_hh), ptr(db_hh), ptr(dw_dec), ptr(db_dec), b, h, stream())


def ce_loss(pred, label, loss_out, ws, *, rows, C, ld, HW, class_weight=None, vis_mask=None, pix_weight=None, pw_bstride=0,
            denom=None, denom_eps=0.0, weight=1.0, dpred=None, smoothing=0.0, focal_gamma=-1.0):
  lib.tfpp_ce_loss(ptr(pred), ptr(label), ptr(class_weight), ptr(vis_mask), ptr(pix_weight), pw_bstride, HW, ptr(denom), denom_eps,
                   weight, ptr(loss_out), ptr(dpred), ptr(ws), ptr(gridsum_scratch(pred.device)), rows, C, ld, float(smoothing), float(focal_gamma),
                   dt(pred), stream())


def reg_loss(pred, target, loss_out, *, B, C, HW, ld, kind, elem_weight=None, wC=1, w_bcast=False, denom=None, denom_eps=0.0,
             denom_mul=1.0, weight=1.0, dpred=None):
  lib.tfpp_reg_loss(ptr(pred), ptr(target), ptr(elem_weight), wC, int(w_bcast), ptr(denom), denom_eps, denom_mul, weight,
                    ptr(loss_out), ptr(dpred), ptr(gridsum_scratch(pred.device)), B, C, HW, ld, kind, dt(pred), stream())


def min_l1_pair_loss(pair, label, loss_out, sel_label, weight=1.0, dpair=None):
  """pair [B, 2, n...] fp32, label [B, n...]: the two-hypothesis waypoint loss of config.multi_wp_output (model.py:401-408); sel_label [B] fp32 <- arg-min."""
  b = pair.shape[0]
  lib.tfpp_min_l1_pair_loss(ptr(_chk(pair)), ptr(_chk(label)), weight, ptr(loss_out), ptr(dpair), ptr(sel_label), b, pair.numel() // (2 * b), stream())


def bce_logits_loss(logit, y, loss_out, weight=1.0, dlogit=None):
  """logit [B, ld] fp32 (channel 0 real), y [B] fp32: nn.BCEWithLogitsLoss() (model.py:266-267,409-411)."""
  lib.tfpp_bce_logits_loss(ptr(_chk(logit)), logit.shape[1], ptr(y), weight, ptr(loss_out), ptr(dlogit), logit.shape[0], stream())


def adamw_amsgrad(p, g, m, v, vmax, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, no_decay_bits=None):
  """no_decay_bits (int32 device tensor from no_decay_bitmask; p must start at the arena's first element): the weight_decay = 0 group of
  create_optimizer_groups."""
  if no_decay_bits is not None:
    lib.tfpp_adamw_amsgrad_groups(ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), p.numel(), lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                                  ptr(no_decay_bits), stream())
    return
  lib.tfpp_adamw_amsgrad(ptr(p), ptr(g), ptr(m), ptr(v), ptr(vmax), p.numel(), lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                         stream())


def no_decay_bitmask(slices, no_decay_ids, total):
  """Host side of tfpp_adamw_amsgrad_groups: slices = [(offset, numel, id(param))] of an arena of ``total`` elements whose parameters all start
  on multiples of 4; returns the int32 words (numpy) with one bit per group of 4 elements, set for the parameters in ``no_decay_ids`` (their
  padding elements included)."""
  import numpy as np
  groups = (total + 3) // 4
  bits = np.zeros((groups + 31) // 32 * 32, dtype=bool)
  for off, n, pid in slices:
    assert off % 4 == 0
    if pid in no_decay_ids:
      bits[off // 4:(off + n + 3) // 4] = True
  words = np.packbits(bits.reshape(-1, 32), axis=1, bitorder='little').view('<u4').reshape(-1)
  return words.astype(np.uint32).view(np.int32)
