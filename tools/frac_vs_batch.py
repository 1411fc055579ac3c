#!/usr/bin/env python3
"""Roofline-vs-size diagnostic: bench.py --batch {32,64,128,256} at M=8, T=64 in fp32 and bf16 -> one JSON document with, per batch
size: clips/s, step times, and the roofline entry of the decoder unit (the chained launch while B*M <= 256 workgroups are
resident at once, else the per-block launches)."""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for prec in ('fp32', 'bf16'):
  for B in (32, 64, 128, 256):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--batch', str(B), '--precision', prec, '--steps', '10', '--warmup', '3',
           '--no-cpu-baseline', '--no-bf16-extra']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
      out['%s_B%d' % (prec, B)] = {'error': r.stderr[-400:]}
      continue
    d = json.loads(line[-1])
    roof = d.get('roofline') or {}
    keep = {k: roof.get(k) for k in ('label', 'avg_us', 'frac', 'frac_mfma', 'frac_hbm', 'achieved_tflops', 'achieved_gbs', 'traffic')}
    for sub in ('block', 'decoder_segment', 'eval_launch'):
      if isinstance(roof.get(sub), dict):
        keep[sub] = {k: roof[sub].get(k) for k in ('us', 'avg_us', 'frac', 'frac_mfma', 'frac_hbm', 'label')}
    out['%s_B%d' % (prec, B)] = dict(value=d['value'], g_step_ms=d.get('g_step_ms'), d_step_ms=d.get('d_step_ms'),
                                     value_blend_50_50=d.get('value_blend_50_50'), roofline=keep)
    sys.stderr.write('%s B=%d: %s clips/s\n' % (prec, B, d['value']))
print(json.dumps(out, indent=1))
