"""GPU: 1-D k3 conv, 1024 workgroups (B=256, T=64, Cout=256), sweep Cin -> does efficiency grow with the K loop length?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.probe_conv import time_conv
for cin in (64, 128, 256, 512, 1024, 2048):
  us, tf = time_conv(256, cin, 256, 64, iters=20)
  print('B256 cin %5d chunks %3d : %8.1f us  %6.2f TF' % (cin, cin // 16, us, tf))
for B in (128, 256, 512, 1024):
  us, tf = time_conv(B, 256, 256, 64, iters=10)
  print('cin256 B %5d wgs %5d : %8.1f us  %6.2f TF' % (B, B * 4, us, tf))
