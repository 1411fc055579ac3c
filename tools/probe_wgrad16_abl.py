"""GPU: ablations of the 16-bit weight-gradient kernel on the decoder / an audio-encoder shape (results meaningless, times only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, ops16, _lib
from mix_stage_amd._lib import MS_BF16
DEV = 'cuda:0'
L = _lib.lib()
SHAPES = [('dec g8 256->256 k3 T64', 1, 32, 256, 256, 8, 3, 1, 1, 1, 64), ('ae2 64->128 3x3 (32,64)', 2, 32, 64, 128, 1, 3, 1, 1, 32, 64)]
for name, nd, B, cin, cout, groups, k, s, p, H, W in SHAPES:
  sp = (H, W) if nd == 2 else (W,)
  kt = (k, k) if nd == 2 else (k,)
  x = torch.randn((B, cin * groups) + sp, device=DEV)
  w = (torch.randn((cout * groups, cin) + kt, device=DEV) * 0.05).requires_grad_()
  b = torch.zeros(cout * groups, device=DEV, requires_grad=True)
  xc = ops16.to_cb8(x, MS_BF16).detach()
  for target in (128, 512):
    for label, flags in (('full', 0), ('no stores', 0x10), ('no MFMA', 0x20), ('no MFMA, no stores', 0x30), ('no tile loop', 0x40), ('no tile loop, no stores', 0x50)):
      L.ms_debug_set_wgrad16_target(target); L.ms_debug_set_conv16_ring(0, flags)
      geom = ops.ConvGeom(nd, groups, k, s, p)
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
      for r in rows:
        if 'wgrad' in r['label'] and 'reduce' not in r['label']:
          print('%-26s target %4d %-26s %7.1f us' % (name, target, label, r['total_ms'] / r['count'] * 1e3))
L.ms_debug_set_conv16_ring(0, 0)
