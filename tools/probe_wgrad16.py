"""GPU: the 16-bit weight-gradient kernel alone on a few layer shapes, against its workgroup target (pixel splits) and ring depth.
  python tools/probe_wgrad16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, ops16, _lib
from mix_stage_amd._lib import MS_BF16
DEV = 'cuda:0'
L = _lib.lib()
SHAPES = [
    ('dec g8 256->256 k3 T64', 1, 32, 256, 256, 8, 3, 1, 1, 1, 64),
    ('unet 256->256 k3 T64', 1, 32, 256, 256, 1, 3, 1, 1, 1, 64),
    ('ae2 64->128 3x3 (32,64)', 2, 32, 64, 128, 1, 3, 1, 1, 32, 64),
    ('ae4 128->256 3x3 (16,32)', 2, 32, 128, 256, 1, 3, 1, 1, 16, 32),
    ('ae3 128->128 4x4s2 (32,64)', 2, 32, 128, 128, 1, 4, 2, 1, 32, 64),
    ('ae0 1->64 3x3 (64,128)', 2, 32, 1, 64, 1, 3, 1, 1, 64, 128),
]
ONLY = os.environ.get('PROBE_ONLY')
CFGS = ((2, 128),) if os.environ.get('PROBE_ONE_CFG') else ((2, 128), (2, 256), (2, 512), (2, 1024))
for name, nd, B, cin, cout, groups, k, s, p, H, W in SHAPES:
  if ONLY and ONLY not in name:
    continue
  sp = (H, W) if nd == 2 else (W,)
  kt = (k, k) if nd == 2 else (k,)
  x = torch.randn((B, cin * groups) + sp, device=DEV)
  w = (torch.randn((cout * groups, cin) + kt, device=DEV) * 0.05).requires_grad_()
  b = torch.zeros(cout * groups, device=DEV, requires_grad=True)
  geom0 = (nd, groups, k, s, p)
  xc = ops16.to_cb8(x, MS_BF16).detach()
  for ring, target in CFGS:
    L.ms_debug_set_wgrad16_target(target)
    geom = ops.ConvGeom(*geom0)
    def run():
      w.grad = None
      y = ops16.conv_block16(xc, w, b, geom, 0)
      y.backward(torch.ones_like(y))
    for _ in range(2):
      run()
    torch.cuda.synchronize()
    ops.timing_enable(True)
    for _ in range(5):
      run()
    torch.cuda.synchronize()
    rows = ops.timing_report(); ops.timing_enable(False)
    wg = [r for r in rows if 'wgrad' in r['label'] and 'reduce' not in r['label']]
    rd = [r for r in rows if 'reduce' in r['label']]
    for r in wg:
      avg = r['total_ms'] / r['count'] * 1e3
      ravg = sum(q['total_ms'] / q['count'] for q in rd) * 1e3
      print('%-28s ring %d target %4d: %7.1f us %6.1f TF  (+ slab reduce %5.1f us)  %s' % (name, ring, target, avg, r['flops'] / avg / 1e6, ravg, r['label'].split('|')[-1][-40:]))
