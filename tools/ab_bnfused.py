"""In-launch BatchNorm on / off inside the captured train step, same process, alternating (argument: bf16 | fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd import _lib
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
dev = torch.device('cuda:0')
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
L = _lib.lib()
def measure(fused, skip='', reps=40, minw=0, precision='bf16'):
  L.ms_debug_set_bn_fused(fused); L.ms_debug_set_bn_fused_min_workgroups(minw)
  L.ms_debug_set_skip(skip.encode() if skip else None)
  model = bench.build_model(dev, precision)
  ts = MixStageTrainStep(model, use_graphs=True)
  out = {}
  for kind in 'GD':
    for _ in range(4):
      ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
      ts.step(*batch, kind=kind)
    torch.cuda.synchronize()
    out[kind] = (time.perf_counter() - t0) / reps * 1e3
  L.ms_debug_set_skip(None)
  del ts, model
  return out
PREC = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
for rnd in range(3):
  for fused, minw in ((0, 0), (1, 0)) + (((1, 300),) if PREC == 'bf16' else ()):
    m = measure(fused, minw=minw, precision=PREC)
    print('%s round %d fused=%d minw=%d  G %.3f D %.3f' % (PREC, rnd, fused, minw, m['G'], m['D']), flush=True)
