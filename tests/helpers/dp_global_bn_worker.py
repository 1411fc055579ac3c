"""Worker of tests/test_gpu_dp.py::test_global_bn_dp_equals_single_device: one data-parallel rank (gloo, all ranks on cuda:0)
with bn_sync='global'.  Rank r trains on clips [r*Bl, (r+1)*Bl) of a global batch; prints one JSON line."""
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
  M = S = 2
  Bl = 2
  torch.manual_seed(77)
  model = build_hip_gan(M, S)
  if os.environ.get('DP_PRECISION', 'fp32') != 'fp32':
    import mix_stage_amd as A
    A.set_compute_dtype(model, os.environ['DP_PRECISION'])
  ts = MixStageTrainStep(model, use_graphs=True, bn_sync='global')
  assert ts.use_graphs is False
  audio, pose, labels, style = O.synthetic_batch(Bl * world, M=M, S=S, seed=321)
  sl = slice(rank * Bl, (rank + 1) * Bl)
  out = {}
  for kind in ('G', 'D'):
    ts.step(audio[sl].cuda(), labels[sl].cuda(), pose[sl].cuda(), style[sl].contiguous().cuda(), kind=kind)
    out[kind] = dict(losses=[float(l.detach()) for l in ts.losses], pose=ts.fake_pose.detach().cpu().flatten().tolist())
  torch.cuda.synchronize()
  sd = model.state_dict()
  probe = {k: float(sd[k].double().sum()) for k in ('G.decoder.1.conv.weight', 'G.audio_encoder.conv.3.norm.running_var',
                                                    'D.conv3.conv.weight', 'G.unet.conv1.2.norm.weight')}
  _emit(dict(rank=rank, out=out, probe=probe))
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
