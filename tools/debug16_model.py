"""GPU debug: per-parameter gradient agreement of the bf16 mode against the fp64 oracle (one G-step / D-step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import mixstage_oracle as O
from test_gpu_model16 import _hip_gan, _step
M = S = 4; B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kind = sys.argv[2] if len(sys.argv) > 2 else 'G'
batch = O.synthetic_batch(B, M=M, S=S)
b64 = [t.double() if t.is_floating_point() else t for t in batch]
ref = O.build_gan(M=M, S=S, dtype=torch.float64)
hip = _hip_gan(M, S)
acts_r, acts_h = {}, {}
import mix_stage_amd as A
from mix_stage_amd import ops16
for name, m in ref.named_modules():
  if isinstance(m, O.ConvNormRelu):
    def rhook(mod, i, o, n=name):
      acts_r.setdefault(n, o.detach())
    m.register_forward_hook(rhook)
for name, m in hip.named_modules():
  if isinstance(m, A.ConvNormRelu):
    def hook(mod, i, o, n=name):
      c = mod.conv.weight.shape[0]
      acts_h.setdefault(n, (ops16.from_cb8(o.detach(), c) if ops16.is_cb8(o) else o.detach()).cpu())
    m.register_forward_hook(hook)
f_ref, l_ref = _step(ref, b64, kind, 'cpu')
f_hip, l_hip = _step(hip, batch, kind, 'cuda:0')
print('pose l1', (f_hip.cpu().double() - f_ref).abs().mean().item(), 'losses', l_hip, l_ref)
for n in acts_r:
  a, b = acts_h[n].double(), acts_r[n]
  print('act %-40s rel l2 %.4f  flips %.4f' % (n, ((a - b).norm() / b.norm()).item(), ((a > 0) != (b > 0)).float().mean().item()))
none_h = [n for n, p in hip.named_parameters() if p.grad is None]
none_r = [n for n, p in ref.named_parameters() if p.grad is None]
print('HIP params without grad: %d, oracle: %d' % (len(none_h), len(none_r)))
print('  only HIP:', [n for n in none_h if n not in none_r][:12])
for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
  if q.grad is None or p.grad is None:
    continue
  a, b = p.grad.cpu().double(), q.grad
  print('grad %-50s |ref| %.3e rel l2 %.4f cos %.4f' % (n, b.norm().item(), ((a - b).norm() / (b.norm() + 1e-30)).item(),
        ((a * b).sum() / (a.norm() * b.norm() + 1e-30)).item()))
