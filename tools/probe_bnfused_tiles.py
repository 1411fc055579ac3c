"""GPU: the bf16 decoder block (conv + BatchNorm + LeakyReLU in one launch) with the conv tile forced: per-block time inside a
captured chain and which form the launcher took.  python tools/probe_bnfused_tiles.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mix_stage_amd as A
from mix_stage_amd import layers, ops, ops16, _lib
from mix_stage_amd._lib import MS_BF16

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = 'cuda:0'
L = _lib.lib()


def chain(cin, cout, groups, B, T, tile, fused=True):
  torch.manual_seed(0)
  blk = layers.ConvNormRelu(cin // groups if groups > 1 else cin, cout // groups if groups > 1 else cout, type='1d',
                            leaky=True, downsample=False, groups=groups).to(dev).train()
  A.set_compute_dtype(blk, 'bf16')
  x = ops16.to_cb8(torch.randn(B, cin, T, device=dev), MS_BF16)
  L.ms_debug_set_bn_fused(1 if fused else 0)
  L.ms_debug_set_conv16_tile(*tile)
  ops.timing_enable(True)
  with torch.no_grad():
    y = blk(x)
    torch.cuda.synchronize()
    labels = [r['label'] for r in ops.timing_report()]
    ops.timing_enable(False)
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
  L.ms_debug_set_conv16_tile(0, 0)
  L.ms_debug_set_bn_fused(1)
  return dt, labels


for name, (cin, cout, groups, B, T) in {
    'decoder 2048->2048 g8 k3, 2048 px': (2048, 2048, 8, 32, 64),
    'decoder M=4 1024->1024 g4 k3': (1024, 1024, 4, 32, 64),
    'unet 256->256 k3, 2048 px': (256, 256, 1, 32, 64),
}.items():
  for tile in ((0, 0), (1, 2), (2, 2), (2, 1), (1, 1)):
    for fused in (True, False):
      try:
        dt, labels = chain(cin, cout, groups, B, T, tile, fused)
        print('%-34s tile %s fused=%d  %.2f us per block   %s' % (name, tile, fused, dt, ' | '.join(l[-60:] for l in labels)), flush=True)
      except Exception as e:
        print('%-34s tile %s fused=%d  failed: %s' % (name, tile, fused, str(e)[:100]), flush=True)
