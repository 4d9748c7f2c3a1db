"""GPU tests of the data-parallel entry (-m gpu): the RCCL path is executed for real with a one-rank process group (a gpurun box has
one GPU; the 2/4/8-GPU curve is the driver's to measure), and bench.py refuses to report an N-GPU number from fewer devices."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def test_one_rank_rccl_group_runs_the_bucketed_exchange_eager_and_behind_one_hipgraph():
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dist_gpu_worker.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                     timeout=600, check=False)
  text = p.stdout.decode()
  assert p.returncode == 0, text[-3000:]
  line = [l for l in text.splitlines() if l.startswith('RESULT ')][-1]
  r = json.loads(line[len('RESULT '):])
  try:
    with open(os.path.join(ROOT, 'gpurun_out', 'model_report.jsonl'), 'a', encoding='utf-8') as f:
      f.write(json.dumps({'test': 'rccl_one_rank', 'errs': r}) + '\n')
  except OSError:
    pass
  assert r['backend'] == 'nccl' and r['world'] == 1
  assert r['calls_local'] == 0                                   # the local step issues no collective
  assert r['layout_agreements'] == 2                             # the ranks compare the arena layout once per layout (static, then observed)
  assert r['buckets_static'] == 2 and r['calls_eager_step'] == 2 and r['async_eager_step'] == 2  # before a pass has been observed: the two static buckets
  assert r['bytes_eager_step'] == r['arena_bytes']               # together exactly one pass over the 481 MB arena
  # the captured step: >= 4 buckets in completion order, every one all-reduced once, all but the last behind an in-graph completion signal
  assert r['buckets_observed'] >= 4 and r['calls_graph_step'] == r['buckets_observed'] and r['async_graph_step'] == r['buckets_observed'], r
  assert r['early_signals'] == r['buckets_observed'] - 1 and r['poisoned'] is None and r['wait_timeouts'] == 0, r
  assert r['bytes_graph_step'] == r['arena_bytes_observed'] >= r['arena_bytes']   # one pass over the arena (bucket starts are padded to 128 elements)
  # a one-rank SUM is the identity: the exchanged step must reproduce the local step (fp32, same tolerance as
  # tests/test_model.py::test_streams_and_hipgraph_do_not_change_the_training_step)
  assert r['loss_eager'] < 1e-4 and r['grad_eager'] < 2e-2 and r['param_eager'] < 1e-6, r
  assert r['loss_graph'] < 1e-2 and r['grad_graph'] < 1e-1, r


def _two_ranks(mode):
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), TFPP_FORCE_COLLECTIVES='0')
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_two_ranks_worker.py'), mode], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT))
  res = []
  for p in procs:
    try:
      text = p.communicate(timeout=900)[0].decode()
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
    assert p.returncode == 0, text[-3000:]
    res.append(json.loads([l for l in text.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):]))
  try:
    with open(os.path.join(ROOT, 'gpurun_out', 'model_report.jsonl'), 'a', encoding='utf-8') as f:
      f.write(json.dumps({'test': 'two_ranks_one_gpu_' + mode, 'errs': res}) + '\n')
  except OSError:
    pass
  return res


def test_two_ranks_one_gpu_trainer_gradient_sum_layout_agreement_and_bit_equal_replicas():
  """the REAL Trainer with two ranks (two processes sharing this box's GPU, 'gloo' on device tensors), different batches per
  rank (train.py:544-553).  The exchanged arena is the sum of the two local gradients; each rank observes its own backward pass and the layouts
  agree; after three eager steps and two replays of the captured step the replicas are bit-equal and no completion-signal wait gave up."""
  for r in _two_ranks('trainer'):
    assert r['world'] == 2 and r['backend'] == 'gloo'
    assert r['grad_sum_rel'] < 1e-5 and r['ranks_see_the_same_sum'], r         # (gloo sums on the host in fp32: not the kernels' order)
    assert r['layout_final_after_eager'] and r['layouts_equal'] and r['buckets'] >= 3 and r['poisoned'] is None, r
    assert r['steps'] == 5 and r['wait_timeouts'] == 0 and r['params_finite'] and r['replicas_bit_equal'], r
    assert r['losses_differ_between_ranks'], r                               # the ranks really trained on different batches


def test_two_ranks_one_gpu_dropin_module_under_distributed_data_parallel():
  """The drop-in module wrapped by torch's DistributedDataParallel as train.py:516-520 wraps the reference's, two ranks on one GPU ('gloo'),
  different batches, five steps of the restated train.py loop (the last two replayed from hipGraphs): replicas bit-equal, no wait gave up."""
  for r in _two_ranks('dropin'):
    assert r['world'] == 2 and r['graph_steps'] >= 2 and r['buckets'] >= 3, r
    assert r['wait_timeouts'] == 0 and r['params_finite'] and r['replicas_bit_equal'] and r['losses_differ_between_ranks'], r


def test_two_ranks_one_gpu_sync_batchnorm_equals_plain_batchnorm_over_the_joint_batch():
  """train.py:511-512 (config.sync_batch_norm = 1): a module converted by nn.SyncBatchNorm.convert_sync_batchnorm, two ranks with two samples each
  ('gloo' on one GPU), against plain BatchNorm over the same four samples in one process: forward rows, BatchNorm running statistics, and the sum
  over the ranks of the parameter gradients (identical per-sample output gradients as seeds) agree to fp32 summation noise."""
  for r in _two_ranks('syncbn'):
    assert r['world'] == 2
    assert max(r['fwd_rel'].values()) < 1e-4, r
    # gradients of this network at four samples per BatchNorm are ill-conditioned in fp32 (the one-rank exchange test above allows
    # 2e-2 between two runs that differ only in the order of fp32 sums): whole-arena distance within 5e-2, per-tensor norms tight in the median
    st = r['grad_stats']
    assert r['running_stats_rel'] < 1e-5 and r['grad_sum_rel'] < 5e-2 and r['own_grad_differs_from_sum'], r
    assert st['arena_cosine'] > 0.998 and st['norm_err_median'] < 5e-3 and st['norm_err_p90'] < 3e-2, r


def test_bench_refuses_to_report_more_gpus_than_it_runs_on():
  """`python bench.py --gpus 2` without a launcher used to run one rank and print n_gpus: 1."""
  import torch
  if torch.cuda.device_count() >= 2:
    pytest.skip('this box really has >= 2 GPUs')
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, check=False)
  assert p.returncode != 0
  assert 'refusing' in p.stdout.decode() and '"n_gpus"' not in p.stdout.decode()
  env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')  # a launcher that started fewer ranks than --gpus says
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, check=False)
  assert p.returncode != 0 and 'WORLD_SIZE=1' in p.stdout.decode()
