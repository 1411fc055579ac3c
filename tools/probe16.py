"""GPU: timing probe of the 16-bit conv block kernels (forward, data gradient, weight gradient) on the path's main shapes.
usage: probe16.py [wm wn]   (force a forward/dgrad tile)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, ops16, _lib
from mix_stage_amd._lib import MS_BF16
DEV = 'cuda:0'
if len(sys.argv) > 2:
  _lib.lib().ms_debug_set_conv16_tile(int(sys.argv[1]), int(sys.argv[2]))
if len(sys.argv) > 3:
  _lib.lib().ms_debug_set_conv16_ring(int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 0)
ONLY = os.environ.get('PROBE_ONLY')
SHAPES = [
    # name, nd, B, cin, cout, groups, k, s, p, H, W
    ('dec g8 256->256 k3 T64', 1, 32, 256, 256, 8, 3, 1, 1, 1, 64),
    ('mid g1 256->256 k3 T64', 1, 32, 256, 256, 1, 3, 1, 1, 1, 64),
    ('down g1 256->256 k4s2 T64', 1, 32, 256, 256, 1, 4, 2, 1, 1, 64),
    ('ae2 64->128 3x3 (32,64)', 2, 32, 64, 128, 1, 3, 1, 1, 32, 64),
    ('ae3 128->128 4x4s2 (32,64)', 2, 32, 128, 128, 1, 4, 2, 1, 32, 64),
    ('ae6 256->256 3x3 (8,16)', 2, 32, 256, 256, 1, 3, 1, 1, 8, 16),
    ('ae7 256->256 3x8 (8,16)', 2, 32, 256, 256, 1, (3, 8), 1, (1, 3), 8, 16),
]
for name, nd, B, cin, cout, groups, k, s, p, H, W in SHAPES:
  if ONLY and ONLY not in name:
    continue
  sp = (H, W) if nd == 2 else (W,)
  kt = (k if isinstance(k, tuple) else (k, k)) if nd == 2 else (k,)
  x = torch.randn((B, cin * groups) + sp, device=DEV).requires_grad_()
  w = (torch.randn((cout * groups, cin) + kt, device=DEV) * 0.05).requires_grad_()
  b = torch.zeros(cout * groups, device=DEV, requires_grad=True)
  g = torch.ones(cout * groups, device=DEV, requires_grad=True); be = torch.zeros(cout * groups, device=DEV, requires_grad=True)
  rm = torch.zeros(cout * groups, device=DEV); rv = torch.ones(cout * groups, device=DEV)
  geom = ops.ConvGeom(nd, groups, k, s, p)
  xc = ops16.to_cb8(x, MS_BF16).detach().requires_grad_()
  def run():
    y = ops16.conv_block16(xc, w, b, geom, 2, gamma=g, beta=be, running_mean=rm, running_var=rv)
    y.backward(torch.ones_like(y))
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  ops.timing_enable(True)
  for _ in range(10):
    run()
  torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  print('== ' + name)
  for r in sorted(rows, key=lambda r: -r['total_ms']):
    avg = r['total_ms'] / r['count'] * 1e3
    print('   %-95s %7.1f us %7.1f TF %6.0f GB/s' % (r['label'].split('|')[-1], avg, r['flops'] / avg / 1e6 if r['flops'] else 0, r['bytes'] / avg / 1e3))
