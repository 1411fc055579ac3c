"""GPU: run ONE conv shape repeatedly (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.probe_conv import time_conv
kind = sys.argv[1] if len(sys.argv) > 1 else 'g8'
if kind == 'g8':
  print(time_conv(32, 256, 256, 64, groups=8, iters=20))
elif kind == 'mid':
  print(time_conv(32, 256, 256, 64, iters=20))
elif kind == '3x3':
  print(time_conv(32, 128, 256, 32, k=3, s=1, p=1, nd=2, H=16, iters=20))
elif kind == 'longk':
  print(time_conv(256, 2048, 256, 64, iters=10))
