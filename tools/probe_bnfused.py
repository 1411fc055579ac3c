"""GPU: what the in-launch BatchNorm costs INSIDE a captured graph: a dependent chain of N decoder / UNet blocks (conv + BN +
LeakyReLU, train mode), replayed; two-launch form against the fused form and the fused form with parts ablated
(ms_debug_set_conv16_ring flag bits 12..15: no y_raw store / no wait / no partial loads / no publish; results meaningless then).
  python tools/probe_bnfused.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mix_stage_amd as A
from mix_stage_amd import layers, ops16, _lib
from mix_stage_amd._lib import MS_BF16

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = 'cuda:0'
L = _lib.lib()


def chain(cin, cout, groups, B, T, fused, flags=0, skip=None):
  torch.manual_seed(0)
  blk = layers.ConvNormRelu(cin // groups if groups > 1 else cin, cout // groups if groups > 1 else cout, type='1d',
                            leaky=True, downsample=False, groups=groups).to(dev).train()
  A.set_compute_dtype(blk, 'bf16')
  x = ops16.to_cb8(torch.randn(B, cin, T, device=dev), MS_BF16)
  L.ms_debug_set_bn_fused(1 if fused else 0)
  L.ms_debug_set_conv16_ring(0, flags)
  L.ms_debug_set_skip(skip)
  with torch.no_grad():
    y = blk(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      y = x
      for _ in range(N):
        y = blk(y)
    for _ in range(3):
      g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      g.replay()
    torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / 20 / N * 1e6
  L.ms_debug_set_conv16_ring(0, 0)
  L.ms_debug_set_skip(None)
  L.ms_debug_set_bn_fused(1)
  ops16._bn_sync[('cuda', 0)].zero_()          # ablated launches leave the counters dirty
  torch.cuda.synchronize()
  return dt


for name, (cin, cout, groups, B, T) in {
    'decoder 2048->2048 g8 k3, 2048 px': (2048, 2048, 8, 32, 64),
    'unet 256->256 k3, 2048 px': (256, 256, 1, 32, 64),
    'unet 256->256 k3, 512 px': (256, 256, 1, 32, 16),
}.items():
  rows = [('two launches (conv+stats, finalize+apply)', dict(fused=False)),
          ('  conv+stats alone (BN launch dropped)', dict(fused=False, skip=b'bn_finalize;bn_apply')),
          ('fused', dict(fused=True)),
          ('  fused, no y_raw store', dict(fused=True, flags=0x1000)),
          ('  fused, no wait', dict(fused=True, flags=0x2000)),
          ('  fused, no wait, no partial loads', dict(fused=True, flags=0x6000)),
          ('  fused, no publish / wait / loads', dict(fused=True, flags=0xE000)),
          ('  fused, none of it and no y_raw', dict(fused=True, flags=0xF000))]
  for label, kw in rows:
    print('%-36s %-44s %.2f us per block' % (name, label, chain(cin, cout, groups, B, T, **kw)))
