"""GPU: K-loop length sweep of the 16-bit grouped decoder conv (t = fixed cost + per-stage cost * stages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, ops16
from mix_stage_amd._lib import MS_BF16
DEV = 'cuda:0'
from mix_stage_amd import _lib
DBG = int(sys.argv[1]) if len(sys.argv) > 1 else 0
_lib.lib().ms_debug_set_conv16_ring(0, DBG << 4)
for cin in (32, 256):
  B, cout, groups, W = 32, 256, 8, 64
  x = torch.randn(B, cin * groups, W, device=DEV)
  w = torch.randn(cout * groups, cin, 3, device=DEV) * 0.05
  b = torch.zeros(cout * groups, device=DEV)
  geom = ops.ConvGeom(1, groups, 3, 1, 1)
  xc = ops16.to_cb8(x, MS_BF16)
  for mode, name in ((0, 'bare'), (2, 'bn_train')):
    g = torch.ones(cout * groups, device=DEV); be = torch.zeros_like(g); rm = torch.zeros_like(g); rv = torch.ones_like(g)
    kw = dict(gamma=g, beta=be, running_mean=rm, running_var=rv) if mode == 2 else {}
    ops.enable_prepared_weights(True)
    with torch.no_grad():
      for _ in range(3):
        ops16.conv_block16(xc, w, b, geom, mode, **kw)
      torch.cuda.synchronize()
      ops.timing_enable(True)
      for _ in range(10):
        ops16.conv_block16(xc, w, b, geom, mode, **kw)
      torch.cuda.synchronize()
    rows = [r for r in ops.timing_report() if 'conv_fwd' in r['label']]
    ops.timing_enable(False)
    r = rows[0]
    print('cin %4d stages %2d %-8s %6.1f us' % (cin, cin // 32, name, r['total_ms'] / r['count'] * 1e3))
