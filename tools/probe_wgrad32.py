"""GPU: fp32 weight-gradient kernel alone on one grouped / plain k3 layer at growing batch (reduction length): the label table
shows the asymptotic rate of the kernel (fixed per-workgroup costs vanish with the reduction length).
  python tools/probe_wgrad32.py [groups] [wave 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import layers, ops, _lib
dev = 'cuda:0'
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wave = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_lib.lib().ms_debug_set_wgrad_wave(wave)
_lib.lib().ms_debug_set_clip32(0)
if len(sys.argv) > 3:
  _lib.lib().ms_debug_set_wgrad_target(int(sys.argv[3]))
torch.manual_seed(0)
for B in (32, 64, 128, 256):
  blk = layers.ConvNormRelu(256, 256, type='1d', leaky=True, downsample=False, groups=groups).to(dev).train()
  x = torch.randn(B, 256 * groups, 64, device=dev, requires_grad=True)
  for it in range(4):
    blk.zero_grad()
    y = blk(x)
    if it == 3:
      torch.cuda.synchronize(); ops.timing_enable(True)
    y.backward(torch.ones_like(y))
  torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  for r in rows:
    if 'wgrad' in r['label'] and 'reduce' not in r['label']:
      avg = r['total_ms'] / r['count'] * 1e3
      print('B=%-4d %-80s %8.1f us %6.1f TF' % (B, r['label'].split('|')[-1][:80], avg, r['flops'] / (avg * 1e-6) / 1e12))
