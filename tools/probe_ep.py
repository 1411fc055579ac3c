"""GPU: decoder-shaped grouped conv, bare vs BN-train epilogue, per-kernel time from the library's HIP-event timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops
from mix_stage_amd._lib import MS_BARE, MS_BN_TRAIN, MS_LRELU

dev = 'cuda:0'
def run(groups, cin, cout, mode, B=32, T=64, iters=20):
  x = torch.randn(B, cin * groups, T, device=dev)
  w = torch.randn(cout * groups, cin, 3, device=dev) * 0.05
  b = torch.zeros(cout * groups, device=dev)
  ga, be = torch.ones(cout * groups, device=dev), torch.zeros(cout * groups, device=dev)
  rm, rv = torch.zeros(cout * groups, device=dev), torch.ones(cout * groups, device=dev)
  geom = ops.ConvGeom(1, groups, 3, 1, 1)
  kw = dict(gamma=ga, beta=be, running_mean=rm, running_var=rv) if mode == MS_BN_TRAIN else {}
  for _ in range(3): ops.conv_block(x, w, b, geom, mode, **kw)
  torch.cuda.synchronize()
  ops.timing_enable(True)
  for _ in range(iters): ops.conv_block(x, w, b, geom, mode, **kw)
  torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  for r in sorted(rows, key=lambda r: -r['total_ms']):
    avg = r['total_ms'] / r['count'] * 1e3
    print('   %-70s x%-3d %8.1f us %6.1f TF' % (r['label'].split('|')[-1], r['count'], avg, r['flops'] / avg / 1e6 if r['flops'] else 0))

for name, mode in (('bare', MS_BARE), ('lrelu', MS_LRELU), ('bn_train', MS_BN_TRAIN)):
  print(name, 'g8'); run(8, 256, 256, mode)
for name, mode in (('bare', MS_BARE), ('bn_train', MS_BN_TRAIN)):
  print(name, 'g1 B=256'); run(1, 256, 256, mode, B=256)
