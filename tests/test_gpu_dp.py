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


def test_two_ranks_stay_bit_identical():
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', '29541', os.path.join(ROOT, 'tests', 'helpers', 'dp_worker.py')]
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  res = [json.loads(l.split('DPRESULT ', 1)[1]) for l in out.stdout.splitlines() if 'DPRESULT ' in l]
  assert out.returncode == 0 and len(res) == 2, (out.stdout[-2000:], out.stderr[-4000:])
  a, b = sorted(res, key=lambda r: r['rank'])
  assert a['kinds'] == b['kinds'] and set(a['kinds']) <= {'G', 'D'}
  assert a['sums'] == b['sums'], (a['sums'], b['sums'])            # bit-identical replicas
  assert (a['g_step'], a['d_step']) == (b['g_step'], b['d_step']) == (a['kinds'].count('G'), a['kinds'].count('D'))
  assert a['losses'] != b['losses']                                # the ranks really saw different shards
  assert all(abs(v) < 1e3 for step in a['losses'] for v in step)
