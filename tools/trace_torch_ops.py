"""GPU: which aten ops (outside the HIP library) run on the device in one G-step and one D-step (eager), with shapes and the
nearest frame of this package.  usage: trace_torch_ops.py [fp32|bf16]"""
import collections
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O

dev = torch.device('cuda:0')
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=False)
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
for k in 'GDGD':
  ts.step(*batch, kind=k)
torch.cuda.synchronize()
SKIP = ('aten.view', 'aten.detach', 'aten.empty', 'aten.as_strided', 'aten.transpose', 'aten.expand', 'aten.unsqueeze', 'aten.select',
        'aten.slice', 'aten.alias', 'aten.squeeze', 'aten.t.', 'aten._unsafe_view', 'aten.permute', 'aten.reshape', 'aten.lift_fresh',
        'aten._local_scalar_dense', 'aten.is_same_size', 'aten.unbind', 'aten.split', 'aten.new_empty', 'aten.resize_')


class Log(TorchDispatchMode):
  def __init__(self):
    super().__init__()
    self.rows = collections.Counter()

  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    name = str(func)
    out = func(*args, **(kwargs or {}))
    if not name.startswith(SKIP):
      tens = [a for a in args if isinstance(a, torch.Tensor)]
      if any(t.is_cuda for t in tens) or (isinstance(out, torch.Tensor) and out.is_cuda):
        frames = [f for f in traceback.extract_stack() if '/mix_stage_amd/' in f.filename]
        where = '%s:%d' % (os.path.basename(frames[-1].filename), frames[-1].lineno) if frames else '(autograd engine)'
        self.rows[(name, tuple(tuple(t.shape) for t in tens[:2]), where)] += 1
    return out


for kind in 'GD':
  with Log() as lg:
    ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
  print('==== %s-step (%s): %d device aten calls' % (kind, precision, sum(lg.rows.values())))
  for (name, shapes, where), n in sorted(lg.rows.items(), key=lambda kv: (-kv[1], kv[0][0])):
    print('  %3d  %-28s %-40s %s' % (n, name, str(shapes)[:40], where))
