"""GPU: N eager (no HIP graph) steps of one kind of the bench workload, for `rocprofv3 --pmc` passes (tools/pmc_step.sh).
  python3 tools/probe_step_eager.py PRECISION KIND N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
precision, kind, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
dev = torch.device('cuda:0')
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=False)
for _ in range(n):
  ts.step(*batch, kind=kind)
torch.cuda.synchronize()
print('ok')
