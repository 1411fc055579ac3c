"""GPU: every launch label of one eager G-step / D-step with its count, average HIP-event duration and rate (the library's
ms_timing_* events, the same source as bench.py's kernel_table, unabridged).  Eager event timing overstates short kernels by
~4 us; use it for WHAT runs, tools/trace_step.py for how long.

  python tools/label_table.py [fp32|bf16] [G|D]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mix_stage_amd.train_step import MixStageTrainStep  # noqa: E402
from oracle import mixstage_oracle as O  # noqa: E402

dev = torch.device('cuda:0')
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
kind = sys.argv[2] if len(sys.argv) > 2 else 'G'
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
model = bench.build_model(dev, precision)
ts = MixStageTrainStep(model, use_graphs=True)
for _ in range(3):
  ts.step(*batch, kind=kind)
roof, rows = bench.kernel_roofline(ts, batch, [kind, kind], precision)
tot = sum(r['total_ms'] for r in rows)
print('%s %s-step: %d labels, %.3f ms of event time over 2 steps' % (precision, kind, len(rows), tot))
for r in rows:
  n = r['count']
  print('%4d x %8.2f us = %7.3f ms  %6.1f TF  %s' % (n // 2, 1e3 * r['total_ms'] / n, r['total_ms'] / 2,
                                                      r['flops'] / (r['total_ms'] / n * 1e-3) / 1e12 if r['flops'] else 0.0, r['label'].split('|')[-1]))
