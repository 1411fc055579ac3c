"""GPU: in-graph time per BN_TRAIN block (conv + BatchNorm + LeakyReLU forward), two-launch form vs in-launch BatchNorm, for the
shape classes of the headline config.  N launches of the block on one input captured in a graph, replayed.
  python tools/probe_bnfused_shapes.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import ops, ops16, _lib
from mix_stage_amd._lib import MS_BF16

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = 'cuda:0'
L = _lib.lib()
B = 32
SHAPES = [
    # name, nd, cin, cout, groups, k, s, p, H, W, in_mode
    ('unet/classify 256->256 k3 T64', 1, 256, 256, 1, 3, 1, 1, 1, 64, 0),
    ('unet.conv1.0 k4s2 T64->32', 1, 256, 256, 1, 4, 2, 1, 1, 64, 0),
    ('unet.conv1.1 k4s2 T32->16', 1, 256, 256, 1, 4, 2, 1, 1, 32, 0),
    ('unet.conv1.2 k4s2 T16->8', 1, 256, 256, 1, 4, 2, 1, 1, 16, 0),
    ('unet.conv1.3 k4s2 T8->4', 1, 256, 256, 1, 4, 2, 1, 1, 8, 0),
    ('unet.conv1.4 k4s2 T4->2', 1, 256, 256, 1, 4, 2, 1, 1, 4, 0),
    ('unet.conv2.0 up2 k3 T4', 1, 256, 256, 1, 3, 1, 1, 1, 4, 2),
    ('unet.conv2.1 up2 k3 T8', 1, 256, 256, 1, 3, 1, 1, 1, 8, 2),
    ('unet.conv2.2 up2 k3 T16', 1, 256, 256, 1, 3, 1, 1, 1, 16, 2),
    ('unet.conv2.3 up2 k3 T32', 1, 256, 256, 1, 3, 1, 1, 1, 32, 2),
    ('unet.conv2.4 up2 k3 T64', 1, 256, 256, 1, 3, 1, 1, 1, 64, 2),
    ('PSE.0 104->64 k3 T64', 1, 104, 64, 1, 3, 1, 1, 1, 64, 0),
    ('PSE.1 64->64 k4s2 T64', 1, 64, 64, 1, 4, 2, 1, 1, 64, 0),
    ('PSE.2 64->128 k4s2 T32', 1, 64, 128, 1, 4, 2, 1, 1, 32, 0),
    ('PSE.3 128->128 k4s2 T16', 1, 128, 128, 1, 4, 2, 1, 1, 16, 0),
    ('PSE.4 128->256 k4s2 T8', 1, 128, 256, 1, 4, 2, 1, 1, 8, 0),
    ('classify.0 266->256 k3 T64', 1, 266, 256, 1, 3, 1, 1, 1, 64, 0),
    ('decoder.0 266->256 g8 bcast', 1, 266, 256, 8, 3, 1, 1, 1, 64, 1),
    ('decoder.1-3 256->256 g8', 1, 256, 256, 8, 3, 1, 1, 1, 64, 0),
    ('D.conv3 128->256 k4 T16->15', 1, 128, 256, 1, 4, 1, 1, 1, 16, 0),
    ('ae.4 128->256 3x3 (16,32)', 2, 128, 256, 1, 3, 1, 1, 16, 32, 0),
    ('ae.5 256->256 4x4s2 ->(8,16)', 2, 256, 256, 1, 4, 2, 1, 16, 32, 0),
    ('ae.6 256->256 3x3 (8,16)', 2, 256, 256, 1, 3, 1, 1, 8, 16, 0),
    ('ae.7 256->256 3x8 (8,16)', 2, 256, 256, 1, (3, 8), 1, (1, 3), 8, 16, 0),
]


def run(shape, fused):
  name, nd, cin, cout, groups, k, s, p, H, W, in_mode = shape
  torch.manual_seed(0)
  ctot = cout * groups
  kt = (k if isinstance(k, tuple) else (k, k)) if nd == 2 else (k,)
  w = torch.randn((ctot, cin) + kt, device=dev) * 0.05
  b = torch.zeros(ctot, device=dev); g = torch.ones(ctot, device=dev); be = torch.zeros(ctot, device=dev)
  rm = torch.zeros(ctot, device=dev); rv = torch.ones(ctot, device=dev)
  cin_tot = cin if in_mode == 1 else cin * groups
  sp = (H, W) if nd == 2 else (W,)
  geom = ops.ConvGeom(nd, groups, k, s, p)
  if in_mode == 2:
    x = ops16.to_cb8(torch.randn((B, cin_tot, W // 2), device=dev), MS_BF16)
    x2 = ops16.to_cb8(torch.randn((B, cin_tot, W), device=dev), MS_BF16)
  else:
    x = ops16.to_cb8(torch.randn((B, cin_tot) + sp, device=dev), MS_BF16)
    x2 = None
  L.ms_debug_set_bn_fused(1 if fused else 0)
  with torch.no_grad():
    def blk():
      return ops16.conv_block16(x, w, b, geom, 2, gamma=g, beta=be, running_mean=rm, running_var=rv, x2=x2, in_mode=in_mode)
    blk()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
      for _ in range(N):
        y = blk()
    for _ in range(3):
      gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
      gr.replay()
    torch.cuda.synchronize()
  L.ms_debug_set_bn_fused(1)
  return (time.perf_counter() - t0) / 20 / N * 1e6


tot = [0.0, 0.0]
for shape in SHAPES:
  u, f = run(shape, False), run(shape, True)
  tot[0] += u; tot[1] += f
  print('%-34s two launches %6.2f us   in-launch BN %6.2f us   %+6.2f' % (shape[0], u, f, f - u))
print('%-34s              %6.2f                   %6.2f' % ('sum', tot[0], tot[1]))
