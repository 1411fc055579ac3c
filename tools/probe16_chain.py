"""GPU: in-graph cost per launch of the 16-bit conv kernel for a few layer geometries and forced tiles: a dependent chain of
N forward calls captured in one HIP graph (BatchNorm kernels dropped through ms_debug_set_skip), replayed.
  python tools/probe16_chain.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mix_stage_amd as A
from mix_stage_amd import layers, ops16, _lib
from mix_stage_amd._lib import MS_BF16

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = 'cuda:0'
L = _lib.lib()


def chain(cin, cout, groups, B, T, tile, skip_bn=True):
  torch.manual_seed(0)
  blk = layers.ConvNormRelu(cin // groups if groups > 1 else cin, cout // groups if groups > 1 else cout, type='1d',
                            leaky=True, downsample=False, groups=groups).to(dev).train()
  A.set_compute_dtype(blk, 'bf16')
  x = ops16.to_cb8(torch.randn(B, cin, T, device=dev), MS_BF16)
  L.ms_debug_set_conv16_tile(*tile)
  L.ms_debug_set_skip(b'bn_finalize;bn_apply' if skip_bn else None)
  with torch.no_grad():
    y = blk(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      y = x
      for _ in range(N):
        y = blk(y) if cin == cout else blk(x)
    for _ in range(3):
      g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      g.replay()
    torch.cuda.synchronize()
  L.ms_debug_set_conv16_tile(0, 0)
  L.ms_debug_set_skip(None)
  return (time.perf_counter() - t0) / 20 / N * 1e6


for name, (cin, cout, groups, B, T) in {
    'decoder 2048->2048 g8 k3, 2048 px': (2048, 2048, 8, 32, 64),
    'unet 256->256 k3, 2048 px': (256, 256, 1, 32, 64),
    'unet 256->256 k3, 512 px': (256, 256, 1, 32, 16),
    'unet 256->256 k3, 128 px': (256, 256, 1, 32, 4),
}.items():
  for tile in ((0, 0), (2, 2), (1, 2), (2, 1), (1, 1)):
    try:
      t = chain(cin, cout, groups, B, T, tile)
      tb = chain(cin, cout, groups, B, T, tile, skip_bn=False) if tile == (0, 0) else float('nan')
      print('%-36s tile %s: %.2f us per conv launch in a chain (with BN kernels: %.2f us per block)' % (name, tile, t, tb))
    except Exception as e:  # noqa: BLE001
      print(name, tile, 'failed', str(e)[:100])
