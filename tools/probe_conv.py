"""GPU probe: time ms_conv_block_fwd (bare conv) over a sweep of shapes with HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops
from mix_stage_amd._lib import MS_BARE, MS_IN_PLAIN

def time_conv(B, cin, cout, T, k=3, s=1, p=1, groups=1, iters=50, nd=1, H=1):
  dev = 'cuda:0'
  if nd == 1:
    x = torch.randn(B, cin * groups, T, device=dev)
    w = torch.randn(cout * groups, cin, k, device=dev) * 0.05
  else:
    x = torch.randn(B, cin * groups, H, T, device=dev)
    w = torch.randn(cout * groups, cin, k, k, device=dev) * 0.05
  b = torch.zeros(cout * groups, device=dev)
  geom = ops.ConvGeom(nd, groups, k, s, p)
  for _ in range(5):
    y = ops.conv_block(x, w, b, geom, MS_BARE)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    y = ops.conv_block(x, w, b, geom, MS_BARE)
  e1.record(); torch.cuda.synchronize()
  us = e0.elapsed_time(e1) / iters * 1e3
  npix = y.numel() // (cout * groups)
  fl = 2.0 * npix * cout * groups * cin * (k if nd == 1 else k * k)
  return us, fl / us / 1e6

if __name__ == '__main__':
  from mix_stage_amd import _lib
  if len(sys.argv) > 1:
    _lib.lib().ms_debug_set_patch_tuning(int(sys.argv[1]), 0)
    print('wide-tile min workgroups set to', sys.argv[1])
  print('--- 1-D k3 s1, M=256, N=2048, sweep Cin (K = 3*Cin)')
  for cin in (8, 21, 64, 128, 256, 512, 1024):
    us, tf = time_conv(32, cin, 256, 64)
    print('cin %5d K %5d : %8.1f us  %6.2f TF' % (cin, 3 * cin, us, tf))
  print('--- 1-D k3 s1, Cin=256, sweep Cout')
  for cout in (64, 128, 256, 512, 1024, 2048):
    us, tf = time_conv(32, 256, cout, 64)
    print('cout %5d : %8.1f us  %6.2f TF' % (cout, us, tf))
  print('--- 1-D k3 s1 256->256, sweep B (N = 64 B)')
  for B in (1, 4, 16, 32, 64, 128, 256):
    us, tf = time_conv(B, 256, 256, 64)
    print('B %5d : %8.1f us  %6.2f TF' % (B, us, tf))
  print('--- grouped decoder g8')
  us, tf = time_conv(32, 256, 256, 64, groups=8); print('g8: %8.1f us %6.2f TF' % (us, tf))
  print('--- 2-D 3x3 128->256 (16,32) B=32 ; 4x4 s2 64->64 (64,128)')
  us, tf = time_conv(32, 128, 256, 32, k=3, s=1, p=1, nd=2, H=16); print('3x3: %8.1f us %6.2f TF' % (us, tf))
  us, tf = time_conv(32, 64, 64, 128, k=4, s=2, p=1, nd=2, H=64); print('4x4s2: %8.1f us %6.2f TF' % (us, tf))
