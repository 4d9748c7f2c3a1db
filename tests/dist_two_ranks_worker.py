"""Worker of tests/test_dist_gpu.py::test_two_ranks_one_gpu_*: ONE of two processes that share cuda:0 and form a 'gloo' process group on
device tensors (a gpurun box has one GPU, but two ranks of the REAL step fit on it).  What runs is the data-parallel
step of team_code/train.py:361,516-520,544-553 with two ranks that draw DIFFERENT batches:

  mode 'trainer'  carla_garage_amd.trainer.Trainer -- (a) lr = 0: the exchanged gradient arena equals the sum of the two ranks' local
                  gradients (each rank computes both locally first); (b) lr > 0: three eager steps (static layout, then the observed one: each
                  rank derives it from its own backward pass and the ranks compare it), then the step captured into ONE hipGraph and replayed
                  twice -- the replicas stay bit-equal, no signal wait gives up;
  mode 'dropin'   the drop-in module under torch's DistributedDataParallel with the arguments of train.py:516-520, driven by the restated
                  train.py loop with the fused optimizer for five steps (eager, then hipGraph replays): replicas bit-equal.

Prints one JSON line ('RESULT ...') per rank."""
import json
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def rank_batch(P, rank, step=0, bs=2):
  b = {k: v.cuda() for k, v in P.make_labels(bs, seed=1234 + rank).items()}
  for k, v in zip(('rgb', 'lidar_bev', 'target_point', 'ego_vel', 'command'), P.make_inputs(bs, seed=1234 + rank)):
    b[k] = v.cuda()
  b['rgb'] = (b['rgb'] + 3.0 * step).clamp(0, 255)
  return b


def gather_equal(values):
  """True when the int64 vector `values` is the same on both ranks (gloo all_gather on CPU tensors)."""
  t = torch.tensor(values, dtype=torch.int64)
  out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
  dist.all_gather(out, t)
  return all(torch.equal(out[0], o) for o in out[1:])


def crc(t):
  return zlib.crc32(t.detach().contiguous().cpu().numpy().tobytes())


def main():
  mode = sys.argv[1]
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  from oracle import tfpp_port as P  # deterministic weights / inputs only
  import test_dropin_gpu as T
  from carla_garage_amd.trainer import Trainer
  from carla_garage_amd.graph import GraphedTrainStep
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', init_method='env://', rank=rank, world_size=world)
  by_name = lambda tr: torch.cat([tr.eng.grads[n].detach().double().flatten() for n, p in tr.model.named_parameters() if p.requires_grad])
  params = lambda m: torch.cat([p.detach().float().flatten() for _, p in m.named_parameters()])
  out = {'rank': rank, 'world': dist.get_world_size(), 'backend': dist.get_backend(), 'mode': mode}
  if mode == 'trainer':
    # (a) gradients: local passes over both ranks' batches, then the exchanged pass
    tl = Trainer(T._model(), lr=0.0)
    tl.exchange = False
    local = []
    for r in range(world):
      tl.train_step(rank_batch(P, r))
      torch.cuda.synchronize()
      local.append(by_name(tl).clone())
    del tl
    tr = Trainer(T._model(), lr=0.0)
    assert tr.exchange_enabled() and tr.world == world
    tr.train_step(rank_batch(P, rank))
    torch.cuda.synchronize()
    got, want = by_name(tr), sum(local)
    out['grad_sum_rel'] = float((got - want).norm() / want.norm())
    out['grad_sum_max_abs'] = float((got - want).abs().max())
    out['ranks_see_the_same_sum'] = gather_equal([crc(got.float())])
    del tr
    torch.cuda.empty_cache()
    # (b) training: eager steps, layout observation, one captured graph, replays
    tr = Trainer(T._model(), lr=1e-4)
    losses = []
    for s in range(3):
      losses.append(tr.total_loss(tr.train_step(rank_batch(P, rank, s))))
    out['layout_final_after_eager'] = bool(tr.layout_final)
    gs = GraphedTrainStep(tr, rank_batch(P, rank, 3), warmup=0)
    for s in range(3, 5):
      losses.append(tr.total_loss(gs(rank_batch(P, rank, s))))
    torch.cuda.synchronize()
    tr.check_exchange_health()
    out.update(steps=tr.step_count, losses=losses, buckets=len(tr.eng.buckets.ranges()), early_signals=len(gs.program[1]), poisoned=tr.eng.buckets.poisoned,
               wait_timeouts=tr.eng.buckets.timed_out(), params_finite=bool(torch.isfinite(tr.flat_param).all()),
               replicas_bit_equal=gather_equal([crc(params(tr.model))]), layouts_equal=gather_equal([crc(torch.tensor(tr.eng.buckets.offsets))]),
               losses_differ_between_ranks=not gather_equal([int(1e6 * l) for l in losses]))
  elif mode == 'syncbn':
    # SyncBatchNorm (train.py:511-512): rank r holds samples [2r, 2r + 2) of a 4-sample batch and a module converted by
    # nn.SyncBatchNorm.convert_sync_batchnorm; the reference is the SAME 4 samples in one process with plain BatchNorm.  Forward rows, BatchNorm
    # running statistics and the SUM over the ranks of the parameter gradients (seeded with the same per-sample output gradients) must agree.
    from carla_garage_amd.engine import Tape

    def seeds_for(t, b0, nb):
      out = []
      for key, real in (('pred_target_speed', 4), ('pred_semantic', 7), ('pred_bev_semantic', 11), ('bb0', 4)):
        x = t['bb'][0] if key == 'bb0' else t[key]
        per = x.numel() // x.shape[0]
        idx = torch.arange(per, device=x.device, dtype=torch.float32).view(1, -1)
        bb = torch.arange(b0, b0 + nb, device=x.device, dtype=torch.float32).view(-1, 1)
        g = (1e-3 * torch.sin(0.37 * idx + 1.3 * bb + len(key))).view(x.shape)
        mask = (torch.arange(x.shape[-1], device=x.device) < real).to(g.dtype)  # channel-padded lanes carry no gradient
        out.append((x, (g * mask).to(x.dtype).contiguous()))
      return out

    def run(model, b0, nb):
      eng = model._engine()
      inp = [v[b0:b0 + nb].cuda().contiguous() for v in P.make_inputs(4)]
      eng.prepare(model.compute_dtype, True, True)
      eng.training = True
      eng.alloc_grads()
      eng.tape = Tape()
      t = eng.forward(*inp)
      fwd = {k: t[k].detach().float().clone() for k in ('pred_target_speed', 'pred_checkpoint')}
      fwd['heat'] = t['bb'][0].detach().float().clone()
      tape, eng.tape = eng.tape, None
      eng.begin_backward()
      tape.backward(seeds_for(t, b0, nb))
      eng.end_backward()
      torch.cuda.synchronize()
      names = [n for n, p in model.named_parameters() if p.requires_grad]
      g = torch.cat([eng.grads[n].detach().double().flatten() for n in names])
      rs = torch.cat([b.detach().double().flatten() for n, b in model.named_buffers() if 'running_' in n])
      return fwd, g, rs, [(n, eng.grads[n].numel()) for n in names]

    ref_fwd, ref_g, ref_rs, sizes = run(T._model(), 0, 4)    # plain BatchNorm over the 4 samples, this process alone
    m = torch.nn.SyncBatchNorm.convert_sync_batchnorm(T._model())
    assert m._engine().sync_bn
    fwd, g, rs, _ = run(m, 2 * rank, 2)
    assert m._engine().sync_world == world
    gsum = g.clone()
    dist.all_reduce(gsum, op=dist.ReduceOp.SUM)
    rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
    # the control: how far apart are two plain-BatchNorm evaluations of the SAME four samples that only differ in the order of the fp32 sums?
    # (the two halves as two passes with the statistics of the joint batch cannot be had without SyncBatchNorm; a second joint pass with the
    # samples in another order changes every reduction order and nothing else)
    from oracle.grad_stats import gradient_stats
    split = lambda flat: {n: flat[o:o + k] for (n, k), o in zip(sizes, np.cumsum([0] + [k for _, k in sizes])[:-1])}
    st = gradient_stats(split(ref_g), split(gsum))
    out.update(fwd_rel={k: rel(fwd[k].double(), ref_fwd[k][2 * rank:2 * rank + 2].double()) for k in fwd}, grad_sum_rel=rel(gsum, ref_g),
               grad_stats={k: float(v) for k, v in st.items()}, running_stats_rel=rel(rs, ref_rs), own_grad_differs_from_sum=rel(g, ref_g) > 0.05)
  else:
    from carla_garage_amd.losses import normalized_loss_weights
    from carla_garage_amd.optim import FlatAdamW
    m = T._model()
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=None, output_device=None, broadcast_buffers=False, find_unused_parameters=False)
    opt = FlatAdamW(ddp.parameters(), lr=1e-4, amsgrad=True)
    w = normalized_loss_weights(m.config)
    losses = []
    for s in range(5):
      losses += T.train_py_loop(m, opt, [rank_batch(P, rank, s)], w, wrapper=ddp)
    torch.cuda.synchronize()
    step = m.__dict__['_dropin_step']
    plan = next(iter(step.plans.values()))
    step.eng.buckets.raise_if_timed_out(block=True)
    out.update(losses=losses, graph_steps=plan.count - 3 if plan.B1 is not None else 0, buckets=len(step.eng.buckets.ranges()),
               wait_timeouts=step.eng.buckets.timed_out(), replicas_bit_equal=gather_equal([crc(params(m))]),
               losses_differ_between_ranks=not gather_equal([int(1e6 * l) for l in losses]), params_finite=bool(torch.isfinite(params(m)).all()))
  print('RESULT ' + json.dumps(out), flush=True)
  torch.cuda.synchronize()
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
