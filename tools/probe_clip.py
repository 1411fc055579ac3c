"""GPU: one clip-resident 1-D block (clip32.hip) launched alone: HIP-event time of N back-to-back launches in a graph, and with
MS_CLIP_DBG=32 the per-workgroup phase stamps (entry / staged / after barrier / K loop done / exchange / arrived / met / end).

  [MS_CLIP_DBG=32] python tools/probe_clip.py [down|up|plain] [cin] [cout]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mix_stage_amd as A  # noqa: E402

dev = 'cuda:0'
kind = sys.argv[1] if len(sys.argv) > 1 else 'plain'
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B, T = 32, 64
torch.manual_seed(0)
m = A.ConvNormRelu(cin, cout, type='1d', leaky=True, downsample=(kind == 'down')).to(dev)
m.train()
x = torch.randn(B, cin, T, device=dev)
with torch.no_grad():
  for _ in range(3):
    y = m(x)
torch.cuda.synchronize()
if os.environ.get('MS_CLIP_DBG'):
  sys.exit(0)
N = 20
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s), torch.no_grad():
  for _ in range(2):
    m(x)
  torch.cuda.synchronize()
  with torch.cuda.graph(g, stream=s):
    for _ in range(N):
      y = m(x)
for _ in range(3):
  g.replay()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
  g.replay()
b.record()
torch.cuda.synchronize()
print('%s %d->%d: %.2f us per block (graph of %d, 10 replays)' % (kind, cin, cout, a.elapsed_time(b) * 1e3 / (10 * N), N))
