"""GPU: the north-star launch alone -- forward of one grouped decoder block (decoder.1-3: k3, 8 groups, 256 -> 256 channels per
group, B*T = 2048 pixels, batch statistics) -- repeated, for rocprofv3 --pmc passes (tools/pmc_decoder.sh).
usage: probe_decoder.py [fp32|bf16x6|bf16] [iters]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if precision == 'bf16x6':
  os.environ['MS_PRECISION'] = 'bf16x6'
import mix_stage_amd as A
from mix_stage_amd import layers
dev = 'cuda:0'
torch.manual_seed(0)
blk = layers.ConvNormRelu(256, 256, type='1d', leaky=True, downsample=False, groups=8).to(dev).train()
x = torch.randn(32, 2048, 64, device=dev)
if precision == 'bf16':
  A.set_compute_dtype(blk, 'bf16')
  from mix_stage_amd import ops16
  from mix_stage_amd._lib import MS_BF16
  x = ops16.to_cb8(x, MS_BF16)
with torch.no_grad():
  for _ in range(iters):
    y = blk(x)
torch.cuda.synchronize()
print('ok', precision, iters)
