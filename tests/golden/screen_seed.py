"""Screens data seeds for the c2r / c3r fixtures (run in the build container only; imports the reference via oracle/refload.py):

    python tests/golden/screen_seed.py c2r_fp32 [first_seed] [count]          # criterion 1: margin of D's pre-activations
    python tests/golden/screen_seed.py robust c2r_fp32 seed [seed ...]         # criterion 2: fp32 vs fp64 gradient norms of a step

For every seed: one D-step forward of the REFERENCE model on synthetic_batch(seed) and the smallest |input| over every LeakyReLU of
the discriminator in that step (fake and real pass), plus the smallest top-2 margin of the pose-style-encoder scores.  The
discriminator's gradients are not a smooth function of its input where a pre-activation sits on the LeakyReLU kink (a 1e-7 change of
the fake pose then moves single gradient elements by percents), so the fixtures use a seed whose margin is >= 1e-4: fp32
implementations that differ in summation order then take the same slopes everywhere in D.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mixstage_oracle as O   # noqa: E402
from oracle import refload                # noqa: E402


def margins(B, T, M, S, dtype, seed, kinds=('D',)):
  audio, pose, labels, style = O.synthetic_batch(B, T=T, M=M, S=S, dtype=dtype, seed=seed)
  out = {}
  for kind in kinds:
    ref = refload.build_ref_gan(M=M, S=S, T=T, dtype=None)
    ref.load_state_dict(O.deterministic_state(ref.state_dict()))
    ref.to(dtype).train()
    seen, pse = [], []
    hooks = [m.register_forward_pre_hook(lambda mod, inp: seen.append(float(inp[0].detach().abs().min())))
             for m in ref.D.modules() if isinstance(m, torch.nn.LeakyReLU)]
    hooks.append(ref.G.pose_style_encoder.register_forward_hook(lambda m, i, o: pse.append(o.detach().double())))
    ref.D_prob = 1.1 if kind == 'D' else -1.0
    torch.manual_seed(7)
    with torch.no_grad():
      ref([audio, labels], pose, **O.model_kwargs(style, T))
    for h in hooks:
      h.remove()
    out[kind] = min(seen)
    if pse:
      top2 = pse[0].topk(2, dim=-1).values
      out['pse'] = float((top2[..., 0] - top2[..., 1]).min())
  return out


def robustness(B, T, M, S, seed, kind):
  """Largest relative difference of any parameter's gradient norm (conv biases in front of BatchNorm aside: exactly 0 in real
  arithmetic) between an fp32 and an fp64 step of the oracle (== the reference, tests/test_oracle_vs_reference.py) on this seed's
  batch: how much the activations that sit within fp32 rounding of a LeakyReLU kink move this draw's gradients."""
  norms = {}
  for dtype in (torch.float32, torch.float64):
    audio, pose, labels, style = O.synthetic_batch(B, T=T, M=M, S=S, seed=seed, dtype=dtype)
    model = O.build_gan(M=M, S=S, T=T, dtype=dtype)
    og = torch.optim.Adam(model.G.parameters(), lr=1e-4)
    od = torch.optim.Adam(model.D.parameters(), lr=1e-4)
    torch.manual_seed(7)
    O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
    norms[dtype] = {n: p.grad.double().norm().item() for n, p in model.named_parameters()
                    if p.grad is not None and not n.endswith('conv.bias')}
  return max((abs(norms[torch.float32][n] - v) / (v + 1e-30), n) for n, v in norms[torch.float64].items())


if __name__ == '__main__':
  sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
  from make_golden import CONFIGS
  if sys.argv[1] == 'robust':
    B, T, M, S = CONFIGS[sys.argv[2]][:4]
    for seed in [int(v) for v in sys.argv[3:]]:
      print(seed, 'G %.2e %s' % robustness(B, T, M, S, seed, 'G'), ' D %.2e %s' % robustness(B, T, M, S, seed, 'D'), flush=True)
    sys.exit(0)
  name = sys.argv[1]
  first = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
  count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
  B, T, M, S, dtype = CONFIGS[name][:5]
  best = []
  for seed in range(first, first + count):
    m = margins(B, T, M, S, dtype, seed, kinds=('D', 'G'))
    best.append((min(m['D'], m['G']), seed, m))
    print(seed, {k: '%.3g' % v for k, v in m.items()}, flush=True)
  best.sort(reverse=True)
  print('best:', best[:5])
