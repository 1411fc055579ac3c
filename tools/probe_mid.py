"""GPU: mid-layer shaped conv (B=32, 256->256, T=64, k3) fwd+bwd per-kernel times for split-K factors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, _lib
from mix_stage_amd._lib import MS_BN_TRAIN
dev = 'cuda:0'
def run(sk, B=32, C=256, T=64, iters=20):
  _lib.lib().ms_debug_set_patch_tuning(0, sk)
  x = torch.randn(B, C, T, device=dev, requires_grad=True)
  w = (torch.randn(C, C, 3, device=dev) * 0.05).requires_grad_()
  b = torch.zeros(C, device=dev, requires_grad=True)
  ga, be = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
  rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
  geom = ops.ConvGeom(1, 1, 3, 1, 1)
  def step():
    y = ops.conv_block(x, w, b, geom, MS_BN_TRAIN, gamma=ga, beta=be, running_mean=rm, running_var=rv)
    y.backward(torch.ones_like(y))
  for _ in range(3): step()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(10): step()
  g.replay(); torch.cuda.synchronize()
  e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
  print('force_splitk %d: graph replay %.1f us per fwd+bwd' % (sk, e0.elapsed_time(e1) * 100))
  ops.timing_enable(True)
  for _ in range(iters): step()
  torch.cuda.synchronize()
  rows = ops.timing_report(); ops.timing_enable(False)
  for r in sorted(rows, key=lambda r: -r['total_ms']):
    print('   %-72s x%-3d %7.1f us' % (r['label'].split('|')[-1], r['count'], r['total_ms'] / r['count'] * 1e3))
for sk in (0, 1, 2, 4):
  run(sk)
