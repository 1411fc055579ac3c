"""GPU: per-kernel-label time table of one G-step + one D-step (eager, HIP-event timed by the library)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mix_stage_amd import ops
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
dev = torch.device('cuda:0')
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=False)
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
for k in 'GDGD': ts.step(*batch, kind=k)
for kind in 'GD':
  torch.cuda.synchronize()
  ops.timing_enable(True)
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  ts.step(*batch, kind=kind)
  e1.record(); torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  rows.sort(key=lambda r: -r['total_ms'])
  tot = sum(r['total_ms'] for r in rows)
  cat = {}; ncat = {}
  for r in rows:
    c = r['label'].split('|')[-1].split()[0]
    cat[c] = cat.get(c, 0) + r['total_ms']
    ncat[c] = ncat.get(c, 0) + r['count']
  print('==== %s-step: eager wall %.2f ms, timed kernels %.2f ms, %d launches' % (kind, e0.elapsed_time(e1), tot, sum(r['count'] for r in rows)))
  print('  by category (ms, launches):', {k: (round(v, 3), ncat[k]) for k, v in sorted(cat.items(), key=lambda kv: -kv[1])})
  for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    avg = r['total_ms'] / r['count'] * 1e3
    tf = r['flops'] / (avg * 1e-6) / 1e12 if r['flops'] else 0
    gb = r['bytes'] / (avg * 1e-6) / 1e9
    print('  %-62s x%-3d avg %8.1f us  tot %6.3f ms  %6.1f TF %7.0f GB/s' % (r['label'].split('|')[-1], r['count'], avg, r['total_ms'], tf, gb))
