"""Worker of tests/test_gpu_dp.py::test_rccl_path_single_rank: ONE rank on the `nccl` backend (= RCCL on ROCm) taking the
data-parallel form of the train step (MS_DP_SINGLE_RANK=1): process-group creation on the device, the parameter broadcast, the
the bucketed RCCL all-reduces (AVG) of the live gradient prefix captured INSIDE the step's HIP graph, the first bucket started at
the backward-pass marker on the communication stream.  With one rank the mean over ranks
is the identity, so the run must reproduce the plain single-process step bit for bit."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch
import torch.distributed as dist

from oracle import mixstage_oracle as O


def run(dp, precision, captured=False):
  from test_gpu_model import build_hip_gan
  import mix_stage_amd as A
  from mix_stage_amd.train_step import MixStageTrainStep
  os.environ['MS_DP_SINGLE_RANK'] = '1' if dp else '0'
  os.environ['MS_CAPTURE_ALLREDUCE'] = '1' if captured else '0'      # default form: two graphs around an eager RCCL exchange
  torch.manual_seed(1234)
  model = build_hip_gan(4, 4)
  if precision != 'fp32':
    A.set_compute_dtype(model, precision)
  ts = MixStageTrainStep(model, use_graphs=True, overlap_allreduce=dp and captured)   # (captured + overlapped: the opt-in form)
  assert (ts.world > 1) == dp
  losses = []
  for i in range(5):
    audio, pose, labels, style = O.synthetic_batch(4, M=4, S=4, seed=100 + 10 * i)
    ts.step(audio.cuda(), labels.cuda(), pose.cuda(), style.cuda(), kind='G' if i % 2 == 0 else 'D')
    losses.append([float(l.detach()) for l in ts.losses])
  torch.cuda.synchronize()
  one_graph = all(e['opt'] is None for e in ts._graphs.values())      # RCCL: the exchange is captured inside the step's graph
  return dict(sums=ts.state_checksums(), losses=losses, one_graph=one_graph, overlap=bool(ts.overlap_allreduce) if dp else None,
              buckets=[list(b) for b in ts._bucket_bounds(ts.optim_G, ts.optim_G.live_elems(ts.optim_G.active_params()))] if dp else None)


def main():
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
  out = {}
  for precision in ('fp32', 'bf16'):
    out[precision] = dict(plain=run(False, precision), dp=run(True, precision), dp_captured=run(True, precision, captured=True))
  rec = dict(rank=0, out=out, backend=dist.get_backend())
  d = os.environ.get('DP_RESULT_DIR')
  if d:
    with open(os.path.join(d, 'rank0.json'), 'w') as f:
      json.dump(rec, f)
  else:
    print('DPRESULT ' + json.dumps(rec), flush=True)
  dist.barrier()
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
