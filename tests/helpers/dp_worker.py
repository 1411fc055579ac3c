"""Worker of tests/test_gpu_dp.py: one data-parallel rank (gloo, all ranks on cuda:0).  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch
import torch.distributed as dist

from oracle import mixstage_oracle as O


def _emit(rec):
  """One record per rank: a file when the test asks for it (two ranks' long lines interleave in a shared pipe), else stdout."""
  d = os.environ.get('DP_RESULT_DIR')
  if d:
    with open(os.path.join(d, 'rank%d.json' % rec['rank']), 'w') as f:
      json.dump(rec, f)
  else:
    print('DPRESULT ' + json.dumps(rec), flush=True)



def main():
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group('gloo', rank=rank, world_size=world)
  torch.cuda.set_device(0)
  from test_gpu_model import build_hip_gan
  from mix_stage_amd.train_step import MixStageTrainStep
  torch.manual_seed(1234)                      # same host generator on every rank: same D/G decisions
  model = build_hip_gan(4, 4)
  if rank == 1:                                # rank 1 starts from different weights: the broadcast must fix that
    with torch.no_grad():
      for p in model.parameters():
        p.mul_(1.5)
  if os.environ.get('DP_PRECISION', 'fp32') != 'fp32':
    import mix_stage_amd as A
    A.set_compute_dtype(model, os.environ['DP_PRECISION'])
  ts = MixStageTrainStep(model, use_graphs=True, grad_exchange=os.environ.get('DP_GRAD_EXCHANGE', 'fp32'))
  from mix_stage_amd import ops16
  kinds = []
  losses = []
  for i in range(6):
    audio, pose, labels, style = O.synthetic_batch(4, M=4, S=4, seed=100 + 10 * i + rank)   # a different shard per rank
    kinds.append(ts.step(audio.cuda(), labels.cuda(), pose.cuda(), style.cuda()))
    losses.append([float(l.detach()) for l in ts.losses])
  torch.cuda.synchronize()
  ts.check_health()                            # raises if an in-launch meeting timed out
  _emit(dict(rank=rank, kinds=kinds, sums=ts.state_checksums(), losses=losses, in_launch=ops16.in_launch_meetings(),
                                      g_step=ts.optim_G.step_count, d_step=ts.optim_D.step_count))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
