import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops
from mix_stage_amd._lib import MS_BN_TRAIN, MS_BARE
dev = 'cuda:0'
def run(cin, cout, k, s, p, T, groups=1, B=4):
  torch.manual_seed(0)
  x = torch.randn(B, cin * groups, T, device=dev, requires_grad=True)
  w = (torch.randn(cout * groups, cin, k, device=dev) * 0.05).requires_grad_()
  b = torch.zeros(cout * groups, device=dev, requires_grad=True)
  geom = ops.ConvGeom(1, groups, k, s, p)
  outs = []
  for on in (False, True, True):
    ops.enable_prepared_weights(on) if not (on and outs and len(outs) == 2) else None
    x.grad = None
    y = ops.conv_block(x, w, b, geom, MS_BARE)
    y.backward(torch.ones_like(y) * 0.5 + y.detach() * 0.1)
    outs.append(x.grad.clone())
    if on and len(outs) == 2:
      with torch.no_grad(): pass
  print(cin, cout, k, s, T, groups, 'entries', len(ops._prepared['entries']), [ (e['n'], e['version'], e['tune']) for e in ops._prepared['entries'].values()],
        'equal', torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]), float((outs[0] - outs[1]).abs().max()))
  ops.enable_prepared_weights(False)
run(256, 256, 4, 2, 1, 64)
run(64, 128, 4, 2, 1, 32)
run(256, 256, 3, 1, 1, 64)
run(266, 256, 3, 1, 1, 64)
run(256, 8, 4, 2, 1, 2)
run(7, 5, 3, 1, 1, 37, groups=3)
