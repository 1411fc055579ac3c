"""GPU: fp32 forward + backward of one grouped decoder block, repeated (for tools/pmc_kernel.sh over the fp32 kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mix_stage_amd import layers
dev = 'cuda:0'
torch.manual_seed(0)
blk = layers.ConvNormRelu(256, 256, type='1d', leaky=True, downsample=False, groups=8).to(dev).train()
x = torch.randn(32, 2048, 64, device=dev, requires_grad=True)
for _ in range(10):
  blk.zero_grad()
  y = blk(x)
  y.backward(torch.ones_like(y))
torch.cuda.synchronize()
print('ok')
