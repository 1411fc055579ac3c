"""GPU: which aten ops (outside the HIP library) launch kernels in one G-step and one D-step (eager)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, collections
import bench
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=False)
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
for k in 'GDGD': ts.step(*batch, kind=k)
torch.cuda.synchronize()
for kind in 'GD':
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
  print('====', kind)
  rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith('aten::') and e.self_device_time_total > 0]
  rows.sort(key=lambda e: -e.count)
  for e in rows[:40]:
    st = [f for f in (e.stack or []) if 'mix_stage_amd' in f or 'gan.py' in f or 'train_step' in f]
    print('  %3d  %-26s %7.1f us  %s' % (e.count, e.key, e.self_device_time_total, st[0].split('/repo/')[-1][:80] if st else (e.stack[0][-60:] if e.stack else '?')))
