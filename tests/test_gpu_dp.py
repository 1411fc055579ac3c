"""GPU: the N>1 path end to end -- two data-parallel ranks (gloo; both on cuda:0, the test boxes have one GPU) through
MixStageTrainStep with captured graphs: the split graph around the all-reduce, the rank-0 broadcast, rank-consistent
D/G decisions.  After 6 steps on different shards the replicas' parameters must be bit-identical."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(cmd, env, timeout=240):
  """Both ranks share one GPU through gloo here (the boxes have one): bounded, with one retry on another port.  Every rank
  leaves its record in a file of its own (long lines of two ranks interleave in a shared stdout pipe)."""
  import tempfile
  for attempt in range(2):
    with tempfile.TemporaryDirectory() as d:
      try:
        out = subprocess.run(cmd, env=dict(env, DP_RESULT_DIR=d), capture_output=True, text=True, timeout=timeout)
      except subprocess.TimeoutExpired:
        if attempt:
          raise
        cmd = [c if not c.isdigit() or int(c) < 29000 else str(int(c) + 7) for c in cmd]
        continue
      res = []
      for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name)) as f:
          res.append(json.load(f))
      return out, res


def test_two_ranks_stay_bit_identical():
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29541', os.path.join(ROOT, 'tests', 'helpers', 'dp_worker.py')]
  out, res = _run_ranks(cmd, env)
  assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
  a, b = sorted(res, key=lambda r: r['rank'])
  assert a['kinds'] == b['kinds'] and set(a['kinds']) <= {'G', 'D'}
  assert a['sums'] == b['sums'], (a['sums'], b['sums'])            # bit-identical replicas
  assert (a['g_step'], a['d_step']) == (b['g_step'], b['d_step']) == (a['kinds'].count('G'), a['kinds'].count('D'))
  assert a['losses'] != b['losses']                                # the ranks really saw different shards
  assert all(abs(v) < 1e3 for step in a['losses'] for v in step)


def test_two_processes_on_one_gpu_in_bf16():
  """Two ranks share cuda:0 in the bf16 mode: a launch whose workgroups meet inside the launch (in-launch BatchNorm, chained
  decoder) assumes it has the device to itself, which two processes on one GPU violate -- the trainer detects the shared device
  and keeps those forms off; the replicas end bit-identical, finite, and no meeting timed out (check_health in the worker)."""
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', DP_PRECISION='bf16')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29557', os.path.join(ROOT, 'tests', 'helpers', 'dp_worker.py')]
  out, res = _run_ranks(cmd, env)
  assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
  a, b = sorted(res, key=lambda r: r['rank'])
  assert a['in_launch'] is False and b['in_launch'] is False
  assert a['kinds'] == b['kinds'] and a['sums'] == b['sums'], (a['sums'], b['sums'])
  assert all(v == v and abs(v) < 1e3 for step in a['losses'] for v in step)


def test_two_ranks_bf16_gradient_exchange():
  """grad_exchange='bf16': the all-reduce moves bf16 values; the replicas still end bit-identical (every rank receives the same
  means), and the parameters differ from the fp32 exchange only by the rounding of the gradients."""
  outs = {}
  for mode, port in (('fp32', '29551'), ('bf16', '29553')):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', DP_GRAD_EXCHANGE=mode)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.join(ROOT, 'tests', 'helpers', 'dp_worker.py')]
    out, res = _run_ranks(cmd, env)
    assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
    a, b = sorted(res, key=lambda r: r['rank'])
    assert a['sums'] == b['sums'], (mode, a['sums'], b['sums'])
    outs[mode] = a
  assert outs['fp32']['kinds'] == outs['bf16']['kinds']
  assert outs['fp32']['sums'] != outs['bf16']['sums']                       # the 16-bit wire is really in use
  for net in ('G', 'D'):
    (s32, n32), (s16, n16) = outs['fp32']['sums'][net], outs['bf16']['sums'][net]
    assert abs(n32 - n16) <= 1e-4 * n32, (net, n32, n16)                    # six Adam steps of 1e-4: the same weights to 1e-4


def test_global_bn_dp_equals_single_device():
  """bn_sync='global': two ranks with 2 clips each reproduce what ONE device computes on the 4 clips (the reference trains on
  one device, layers.py:65-70): poses and losses of a G-step and a D-step against the oracle at B=4, within the fp32 bar."""
  import numpy as np
  import torch
  from oracle import mixstage_oracle as O
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29543', os.path.join(ROOT, 'tests', 'helpers', 'dp_global_bn_worker.py')]
  out, res = _run_ranks(cmd, env)
  assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
  a, b = sorted(res, key=lambda r: r['rank'])
  M = S = 2
  ref = O.build_gan(M=M, S=S)
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  audio, pose, labels, style = O.synthetic_batch(4, M=M, S=S, seed=321)
  for kind in ('G', 'D'):
    torch.manual_seed(77)
    fake, losses, _ = O.oracle_train_step(ref, og, od, audio, pose, labels, style, kind)
    got = np.concatenate([np.array(a['out'][kind]['pose']), np.array(b['out'][kind]['pose'])])
    assert np.abs(got - fake.numpy().reshape(-1)).mean() <= 1e-4, kind
    # each rank's loss terms are means over its own clips: their average over the ranks is the single-device loss
    mean_losses = (np.array(a['out'][kind]['losses']) + np.array(b['out'][kind]['losses'])) / 2
    np.testing.assert_allclose(mean_losses, losses, atol=2e-4)
  sd = ref.state_dict()
  for k, v in a['probe'].items():
    assert b['probe'][k] == v                                            # replicas identical
    assert abs(v - float(sd[k].double().sum())) <= 2e-4 * max(1.0, sd[k].numel() ** 0.5), k


def test_global_bn_dp_bf16_equals_single_device_bf16():
  """bn_sync='global' in the bf16 mode (conv -> fp32 -> BatchNorm over all ranks -> 16 bits): two ranks with 2 clips each
  against ONE device stepping on the 4 clips in bf16 (same weights, same arithmetic mode: the difference is the fp32 detour of
  the normalisation and the summation order of the statistics), and against the fp64 oracle within the bf16 rails."""
  import numpy as np
  import torch
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O
  from test_gpu_model import build_hip_gan
  from mix_stage_amd.train_step import MixStageTrainStep
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0', DP_PRECISION='bf16')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29549', os.path.join(ROOT, 'tests', 'helpers', 'dp_global_bn_worker.py')]
  out, res = _run_ranks(cmd, env)
  assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
  a, b = sorted(res, key=lambda r: r['rank'])
  M = S = 2
  audio, pose, labels, style = O.synthetic_batch(4, M=M, S=S, seed=321)
  ref = O.build_gan(M=M, S=S)
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  torch.manual_seed(77)
  single = build_hip_gan(M, S)
  A.set_compute_dtype(single, 'bf16')
  ts = MixStageTrainStep(single, use_graphs=False)
  for kind in ('G', 'D'):
    torch.manual_seed(77)
    fake64, losses64, _ = O.oracle_train_step(ref, og, od, audio, pose, labels, style, kind)
    ts.step(audio.cuda(), labels.cuda(), pose.cuda(), style.cuda(), kind=kind)
    one = ts.fake_pose.detach().cpu().numpy().reshape(-1)
    got = np.concatenate([np.array(a['out'][kind]['pose']), np.array(b['out'][kind]['pose'])])
    f64 = fake64.numpy().reshape(-1)
    e_dp, e_one, e_pair = np.abs(got - f64).mean(), np.abs(one - f64).mean(), np.abs(got - one).mean()
    print('bf16 global BN %s-step: dp vs fp64 %.4f, single bf16 vs fp64 %.4f, dp vs single %.4f' % (kind, e_dp, e_one, e_pair))
    # both are bf16 roundings of the same fp64 step (different ones: the data-parallel form normalises the fp32 accumulators
    # outside the conv launch): each within the bf16 rail of the oracle, and no further from each other than those two errors
    assert e_dp <= 3e-2 and e_one <= 3e-2, (kind, e_dp, e_one)
    assert e_dp <= 1.5 * e_one + 2e-3, (kind, e_dp, e_one)
    assert e_pair <= e_dp + e_one, (kind, e_pair, e_dp, e_one)
    mean_losses = (np.array(a['out'][kind]['losses']) + np.array(b['out'][kind]['losses'])) / 2
    np.testing.assert_allclose(mean_losses, [float(l) for l in ts.losses], atol=2e-2)
  for k, v in a['probe'].items():
    assert b['probe'][k] == v                                            # replicas identical


def test_rccl_path_single_rank():
  """The RCCL calls of the data-parallel step, executed on the one GPU the box has: a single `nccl` rank forced onto the
  data-parallel form of the step (bucketed all-reduces captured in the step's graph, the first one started at the
  backward-pass marker on the communication stream; broadcast) reproduces the plain step bit for bit,
  in the fp32 and the bf16 mode."""
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
         '--master-port', '29545', os.path.join(ROOT, 'tests', 'helpers', 'dp_rccl1_worker.py')]
  out, res = _run_ranks(cmd, env)
  assert out.returncode == 0 and len(res) == 1, (out.stdout[-2000:], out.stderr[-4000:])
  assert res[0]['backend'] == 'nccl'
  for precision, r in res[0]['out'].items():
    # default form: two graphs per step around an EAGER RCCL all-reduce (the captured form is opt-in until an N > 1 run proves it)
    assert r['plain']['sums'] == r['dp']['sums'], precision
    assert r['plain']['losses'] == r['dp']['losses'], precision
    assert not r['dp']['one_graph'] and not r['dp']['overlap'], r['dp']
    assert len(r['dp']['buckets']) == 1
    # opt-in form (MS_CAPTURE_ALLREDUCE=1 + overlap_allreduce): the exchange inside the step's graph, first bucket at the marker
    assert r['plain']['sums'] == r['dp_captured']['sums'], precision
    assert r['plain']['losses'] == r['dp_captured']['losses'], precision
    assert r['dp_captured']['one_graph'] and r['dp_captured']['overlap'], r['dp_captured']
    b = r['dp_captured']['buckets']
    assert len(b) == 4 and b[0][0] == 0 and sorted(x for lo, hi in b for x in (lo, hi))[-1] == max(hi for lo, hi in b)
    assert sum(hi - lo for lo, hi in b) == max(hi for lo, hi in b)            # the buckets tile the live prefix
