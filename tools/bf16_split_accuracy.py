"""CPU experiment behind DESIGN.md 4b: the whole generator + discriminator forward in train mode with every conv computed
from bf16-split operands (1 term = plain bf16, 3 terms, 6 terms), products exact in fp32, fp32 accumulation, against the fp64
oracle.  Result on this container: fp32 1.2e-6, bf16 1.1e-2, bf16x3 2.2e-5, bf16x6 6.9e-7 (pose L1)."""
import sys, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mixstage_oracle as O
torch.set_num_threads(8)
M = S = 8
B = 8
audio, pose, labels, style = O.synthetic_batch(B, M=M, S=S, seed=3)
def run(mode):
  torch.manual_seed(0)
  gan = O.build_gan(M=M, S=S)
  if mode == 'fp64': gan = gan.double()
  orig1, orig2 = F.conv1d, F.conv2d
  def split(t, terms):
    hi = t.to(torch.bfloat16).to(torch.float32)
    if terms == 1: return [hi]
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    if terms == 2: return [hi, lo]
    lo2 = (t - hi - lo).to(torch.bfloat16).to(torch.float32)
    return [hi, lo, lo2]
  def mk(orig, nterm):
    def f(x, w, b=None, *a, **k):
      if x.dtype != torch.float32: return orig(x, w, b, *a, **k)
      xs, ws = split(x, 2 if nterm == 3 else 3), split(w, 2 if nterm == 3 else 3)
      if nterm == 1:
        return orig(xs[0], ws[0], b, *a, **k)
      if nterm == 3:
        pairs = [(0, 0), (0, 1), (1, 0)]
      else:  # 6 terms
        pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
      y = None
      for i, j in reversed(pairs):       # small terms first
        t = orig(xs[i], ws[j], None, *a, **k)
        y = t if y is None else y + t
      if b is not None: y = y + b.view(1, -1, *([1] * (y.dim() - 2)))
      return y
    return f
  if mode in ('bf16x1', 'bf16x3', 'bf16x6'):
    n = int(mode[-1])
    F.conv1d, F.conv2d = mk(orig1, n), mk(orig2, n)
    torch.nn.functional.conv1d, torch.nn.functional.conv2d = F.conv1d, F.conv2d
  try:
    gan.train()
    gan.D_prob = -1.0
    a = audio.double() if mode == 'fp64' else audio
    p = pose.double() if mode == 'fp64' else pose
    fake, losses, _ = gan([a, labels], p, **O.model_kwargs(style))
  finally:
    F.conv1d, F.conv2d = orig1, orig2
    torch.nn.functional.conv1d, torch.nn.functional.conv2d = orig1, orig2
  return fake.detach().double(), [float(l) for l in losses]
ref, lref = run('fp64')
for mode in ('fp32', 'bf16x1', 'bf16x3', 'bf16x6'):
  out, l = run(mode)
  print('%-7s pose L1 vs fp64 %.3e   max %.3e   losses diff %.2e' % (mode, (out - ref).abs().mean().item(), (out - ref).abs().max().item(), max(abs(a - b) for a, b in zip(l, lref))))
