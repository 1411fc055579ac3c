"""GPU debug probe for the 16-bit conv kernels: simple operands, prints where results differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from mix_stage_amd import ops, ops16
from mix_stage_amd._lib import MS_BF16
DEV = 'cuda:0'
torch.manual_seed(0)

def conv_case(name, B, cin, cout, groups, k, s, p, T, ident=False, out_f32=True, mode=0):
  x = torch.randn(B, cin * groups, T, device=DEV)
  w = torch.randn(cout * groups, cin, k, device=DEV) * (cin * k) ** -0.5
  if ident:
    w.zero_()
    for i in range(min(cout, cin)):
      w[i, i, k // 2] = 1.0
  bias = torch.zeros(cout * groups, device=DEV)
  xc = ops16.to_cb8(x, MS_BF16)
  geom = ops.ConvGeom(1, groups, k, s, p)
  y = ops16.conv_block16(xc, w, bias, geom, mode, out_f32=out_f32)
  if not out_f32:
    y = ops16.from_cb8(y, cout * groups)
  ref = F.conv1d(x.bfloat16().double().cpu(), w.bfloat16().double().cpu(), None, stride=s, padding=p, groups=groups)
  d = (y.double().cpu() - ref).abs()
  print('%-28s max err %.4g (ref max %.3g)' % (name, d.max().item(), ref.abs().max().item()))
  if d.max().item() > 1e-2 * ref.abs().max().item():
    bad = d > 1e-2 * ref.abs().max().item()
    print('   bad fraction %.3f; per-batch %s' % (bad.float().mean().item(), bad.float().mean((1, 2)).tolist()[:8]))
    print('   per-channel(first 40) %s' % [round(v, 2) for v in bad.float().mean((0, 2)).tolist()[:40]])
    print('   per-time(first 40) %s' % [round(v, 2) for v in bad.float().mean((0, 1)).tolist()[:40]])
    print('   y[0,:4,:8]', y[0, :4, :8].tolist())
    print('   ref[0,:4,:8]', ref[0, :4, :8].tolist())

x = torch.randn(2, 16, 8, device=DEV)
c = ops16.to_cb8(x, MS_BF16)
print('roundtrip ok', torch.equal(ops16.from_cb8(c, 16), x.bfloat16().float()))
conv_case('k1 ident 64->64 T64', 2, 64, 64, 1, 1, 1, 0, 64, ident=True)
conv_case('k1 rand 64->64 T64', 2, 64, 64, 1, 1, 1, 0, 64)
conv_case('k1 rand 64->64 T64 cb8out', 2, 64, 64, 1, 1, 1, 0, 64, out_f32=False)
conv_case('k3 ident 64->64 T64', 2, 64, 64, 1, 3, 1, 1, 64, ident=True)
conv_case('k3 rand 64->64 T64', 2, 64, 64, 1, 3, 1, 1, 64)
conv_case('k3 rand 32->64 T64', 2, 32, 64, 1, 3, 1, 1, 64)
conv_case('k3 rand 128->128 T64 B8', 8, 128, 128, 1, 3, 1, 1, 64)
conv_case('k3 rand 128->128 g4 B8', 8, 128, 128, 4, 3, 1, 1, 64)
conv_case('k4s2 rand 64->64 T64', 2, 64, 64, 1, 4, 2, 1, 64)
