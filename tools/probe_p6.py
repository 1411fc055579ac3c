"""GPU: bare 1-D convs (forward, data gradient, weight gradient) in the exact-fp32 mode and in the bf16x6 mode (ms_set_precision)
against an fp64 reference: relative max errors of y, dx, dw."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from mix_stage_amd import ops, _lib
from mix_stage_amd._lib import MS_BARE, MS_BN_TRAIN
L = _lib.lib()
dev = 'cuda:0'
def run(B, cin, cout, T, k, s, p, groups, minwg=0, bwd=True):
  torch.manual_seed(0)
  x = torch.randn(B, cin, T, device=dev, requires_grad=True)
  w = (torch.randn(cout, cin // groups, k, device=dev) * 0.1).requires_grad_()
  b = torch.randn(cout, device=dev, requires_grad=True)
  geom = ops.ConvGeom(1, groups, k, s, p)
  ref = F.conv1d(x.double(), w.double(), b.double(), stride=s, padding=p, groups=groups)
  gy = torch.randn_like(ref)
  gx_ref, gw_ref = torch.autograd.grad(ref, (x, w), gy)
  res = {}
  for mode in (0, 1):
    L.ms_set_precision(mode); L.ms_debug_set_patch_min_workgroups(minwg)
    y = ops.conv_block(x, w, b, geom, MS_BARE)
    e = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    if bwd:
      gx, gw = torch.autograd.grad(y, (x, w), gy.float())
      ex = ((gx.double() - gx_ref).abs().max() / gx_ref.abs().max()).item()
      ew = ((gw.double() - gw_ref).abs().max() / gw_ref.abs().max()).item()
    else: ex = ew = 0
    res[mode] = (e, ex, ew)
  print('B%d cin%d cout%d T%d k%d s%d g%d : fp32 %s   bf16x6 %s' % (B, cin, cout, T, k, s, groups, ['%.1e' % v for v in res[0]], ['%.1e' % v for v in res[1]]))
run(2, 256, 104, 64, 1, 1, 0, 8)
run(2, 256, 104, 64, 1, 1, 0, 1)
run(2, 32, 13, 64, 1, 1, 0, 1)
run(2, 32, 64, 64, 1, 1, 0, 1)
run(2, 64, 64, 64, 3, 1, 1, 1)
run(2, 256, 104, 64, 3, 1, 1, 8)
run(4, 256, 256, 64, 3, 1, 1, 1)
run(4, 64, 128, 32, 4, 2, 1, 1)
run(3, 6, 10, 37, 4, 2, 1, 1)

def run2d(B, cin, cout, H, W, k, s, p):
  torch.manual_seed(0)
  x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
  w = (torch.randn(cout, cin, k, k, device=dev) * 0.1).requires_grad_()
  b = torch.randn(cout, device=dev, requires_grad=True)
  geom = ops.ConvGeom(2, 1, k, s, p)
  ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=p)
  gy = torch.randn_like(ref)
  gx_ref, gw_ref = torch.autograd.grad(ref, (x, w), gy)
  res = {}
  for mode in (0, 1):
    L.ms_set_precision(mode); L.ms_debug_set_patch_min_workgroups(0)
    y = ops.conv_block(x, w, b, geom, MS_BARE)
    gx, gw = torch.autograd.grad(y, (x, w), gy.float())
    res[mode] = [((a.double() - r).abs().max() / r.abs().max()).item() for a, r in ((y, ref), (gx, gx_ref), (gw, gw_ref))]
  print('2d B%d cin%d cout%d %dx%d k%d s%d : fp32 %s   bf16x6 %s' % (B, cin, cout, H, W, k, s, ['%.1e' % v for v in res[0]], ['%.1e' % v for v in res[1]]))
run2d(2, 64, 128, 16, 32, 3, 1, 1)
run2d(2, 5, 7, 9, 37, 3, 1, 1)
run2d(2, 64, 64, 32, 48, 4, 2, 1)
run2d(2, 5, 7, 9, 37, 4, 2, 1)
run2d(2, 256, 256, 16, 32, 4, 2, 1)
