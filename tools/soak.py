"""GPU: soak of the train step -- many steps in graph and eager mode, both arithmetic modes: finite losses, stable memory, no
growth of the deferred-launch bookkeeping.  usage: soak.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd import ops
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O

dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
for precision in ('fp32', 'bf16'):
  for graphs, n in ((True, steps), (False, max(100, steps // 10))):
    torch.manual_seed(4321)
    model = bench.build_model(dev, precision)
    ts = MixStageTrainStep(model, use_graphs=graphs)
    mem = []
    for i in range(n):
      ts.step(*batch)
      if i in (n // 4, n - 1):
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated() >> 20)
    losses = [float(l.detach()) for l in ts.losses]
    ok = all(l == l and abs(l) < 1e4 for l in losses)
    print('%s graphs=%s steps=%d  losses %s finite=%s  MiB at 1/4 and end: %s  keep=%d jobs=%d launches=%d' % (
        precision, graphs, n, [round(l, 4) for l in losses], ok, mem, len(ops._deferred['keep']), len(ops._deferred['jobs']),
        ops._deferred['launches']))
    assert ok and mem[1] <= mem[0] + 8 and not ops._deferred['keep'] and not ops._deferred['jobs']
    del ts, model
    torch.cuda.empty_cache()
from mix_stage_amd import ops16
assert not ops16.bn_sync_error(), 'an in-launch BatchNorm meeting timed out during the soak'
print('soak ok')
