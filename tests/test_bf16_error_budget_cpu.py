"""CPU: where the bf16 mode's pose error comes from.  The fp32 oracle is run at the headline shape
(B=32, T=64, M=S=8, train-mode BatchNorm, audio branch) with the roundings of the 16-bit path emulated one class at a time --
conv weights rounded to bf16, block outputs stored as bf16 (per sub-network and all together), both -- and under torch's own
`autocast(bfloat16)` (what BASELINE.md section 2 quotes for the reference), each against the same forward pass in float64.
The device path's measured figure (tests/test_gpu_model16.py, profiles/r03_precision_report.json: 0.0122) must be explained by
the emulation of exactly its roundings; the report goes to stdout (-s) and, on a GPU box, next to the precision report."""
import json
import os

import pytest
import torch

from oracle import mixstage_oracle as O

B, M, S, T = 32, 8, 8, 64
HIP_MEASURED = 0.0122          # headline bf16 G-step pose L1 vs fp64, profiles/r03_precision_report.json


def _forward(model, batch):
  audio, pose, labels, style = batch
  model.train()
  model.D_prob = -1.0
  with torch.no_grad():
    fake, _, _ = model([audio, labels], pose, **O.model_kwargs(style, T))
  return fake


def _round(t):
  return t.to(torch.bfloat16).to(t.dtype)


def _emulated(batch, weights=False, store=(), inputs=False):
  """fp32 oracle with the 16-bit path's roundings: conv weights (`weights`), the outputs of the conv blocks of the listed
  sub-networks (`store`: the cb8 activations travel as bf16), the network inputs (`inputs`)."""
  model = O.build_gan(M=M, S=S, T=T)
  if weights:
    with torch.no_grad():
      for mod in model.G.modules():
        if isinstance(mod, (torch.nn.Conv1d, torch.nn.Conv2d)):
          mod.weight.copy_(_round(mod.weight))
  hooks = []
  for name, mod in model.G.named_modules():
    if isinstance(mod, O.ConvNormRelu) and any(name.startswith(p) for p in store):
      hooks.append(mod.register_forward_hook(lambda m, i, o: _round(o)))
  if inputs:
    batch = [_round(t) if t.is_floating_point() else t for t in batch]
  try:
    return _forward(model, batch)
  finally:
    for h in hooks:
      h.remove()


@pytest.mark.timeout(600)
def test_bf16_error_budget_of_the_headline_forward():
  torch.manual_seed(0)
  batch = O.synthetic_batch(B, T=T, M=M, S=S)
  ref64 = _forward(O.build_gan(M=M, S=S, T=T, dtype=torch.float64), [t.double() if t.is_floating_point() else t for t in batch])
  err = lambda y: (y.double() - ref64).abs().mean().item()
  subnets = ('audio_encoder', 'unet', 'decoder', 'classify_cluster', 'pose_style_encoder')
  rep = {'fp32 oracle': err(_forward(O.build_gan(M=M, S=S, T=T), batch))}
  rep['conv weights in bf16 only'] = err(_emulated(batch, weights=True))
  for sn in subnets:
    rep['block outputs of %s stored in bf16 only' % sn] = err(_emulated(batch, store=(sn,)))
  rep['all block outputs stored in bf16'] = err(_emulated(batch, store=subnets))
  rep['weights + all block outputs + inputs in bf16 (the roundings of the 16-bit path)'] = err(
      _emulated(batch, weights=True, store=subnets, inputs=True))
  model = O.build_gan(M=M, S=S, T=T)
  with torch.autocast('cpu', dtype=torch.bfloat16):
    rep['torch autocast(bfloat16) of the same forward'] = err(_forward(model, batch).float())
  for k, v in rep.items():
    print('%-86s pose L1 vs fp64 %.5f' % (k, v))
  out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
  if os.path.isdir(out):
    json.dump(rep, open(os.path.join(out, 'bf16_error_budget.json'), 'w'), indent=1)
  full = rep['weights + all block outputs + inputs in bf16 (the roundings of the 16-bit path)']
  assert rep['fp32 oracle'] <= 1e-5
  # the device path's error is the error of its roundings: no kernel adds to it
  assert 0.6 * full <= HIP_MEASURED <= 1.6 * full, (HIP_MEASURED, full)
  # ... and it is not above what torch's own bf16 autocast gives for this forward pass at this shape
  assert HIP_MEASURED <= 1.1 * rep['torch autocast(bfloat16) of the same forward']
  # both classes of rounding matter: neither the weights nor the stored activations alone explain the total
  assert rep['conv weights in bf16 only'] < full and rep['all block outputs stored in bf16'] < full
