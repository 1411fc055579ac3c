"""CPU: the oracle restatement reproduces the golden vectors that tests/golden/make_golden.py
captured from the REFERENCE's own model files (outputs, losses, gradients, post-Adam weights)."""
import os

import numpy as np
import pytest
import torch

from oracle import mixstage_oracle as O

CASES = [('c1_fp32', torch.float32, 2e-5), ('c1_fp64', torch.float64, 1e-11),
         ('c2r_fp32', torch.float32, 2e-5), ('c3r_fp32', torch.float32, 2e-5)]


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize('name,dtype,tol', CASES)
@pytest.mark.parametrize('kind', ['G', 'D'])
def test_train_step_matches_reference_vectors(golden_dir, name, dtype, tol, kind):
  z = _load(golden_dir, name)
  B, T, M, S = [int(v) for v in z['meta']]
  audio, pose = torch.from_numpy(z['audio']), torch.from_numpy(z['pose'])
  labels, style = torch.from_numpy(z['labels']), torch.from_numpy(z['style'])
  # the synthetic generator is part of the fixture contract
  seed = int(z['data_seed']) if 'data_seed' in z.files else 1234       # (screened per fixture: tests/golden/make_golden.py)
  a2, p2, l2, s2 = O.synthetic_batch(B, T=T, M=M, S=S, dtype=dtype, seed=seed)
  assert torch.equal(a2, audio) and torch.equal(p2, pose) and torch.equal(l2, labels) and torch.equal(s2, style)

  model = O.build_gan(M=M, S=S, T=T, dtype=dtype)
  og = torch.optim.Adam(model.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(model.D.parameters(), lr=1e-4)
  torch.manual_seed(7)
  fake, losses, gnorm = O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
  k = kind + '/'
  assert np.abs(fake.numpy() - z[k + 'pose']).mean() <= tol
  np.testing.assert_allclose(losses, z[k + 'losses'], rtol=0, atol=10 * tol)
  np.testing.assert_allclose(gnorm, z[k + 'total_grad_norm'], rtol=1e-4)
  np.testing.assert_allclose(model.G.labels_cap_soft.detach().numpy(), z[k + 'labels_cap_soft'], atol=10 * tol)
  for n, p in model.named_parameters():
    key = k + 'gnorm/' + n
    if p.grad is None:
      assert key not in z.files
      continue
    g = p.grad.double().reshape(-1)
    np.testing.assert_allclose(g.norm().item(), z[key], rtol=1e-3, atol=1e-7)
    stride = max(1, g.numel() // 16)
    np.testing.assert_allclose(g[::stride][:16].numpy(), z[k + 'gsamp/' + n], rtol=1e-3, atol=1e-6)
  mod = model.G if kind == 'G' else model.D
  for n, p in mod.named_parameters():
    np.testing.assert_allclose(p.detach().double().sum().item(), z[k + 'psum/%s.%s' % (kind, n)],
                               rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize('name,dtype,tol', CASES[:1] + CASES[2:3])
def test_eval_and_sample_forward(golden_dir, name, dtype, tol):
  z = _load(golden_dir, name)
  B, T, M, S = [int(v) for v in z['meta']]
  seed = int(z['data_seed']) if 'data_seed' in z.files else 1234
  audio, pose, labels, style = O.synthetic_batch(B, T=T, M=M, S=S, dtype=dtype, seed=seed)
  model = O.build_gan(M=M, S=S, T=T, dtype=dtype).eval()
  with torch.no_grad():
    fake, losses, _ = model([audio, labels], pose, **O.model_kwargs(style, T))
    kw = O.model_kwargs(style, T)
    kw['sample_flag'] = 1
    fake_s, _, _ = model([audio, labels], pose, **kw)
  assert np.abs(fake.numpy() - z['E/pose']).mean() <= tol
  assert np.abs(fake_s.numpy() - z['S/pose']).mean() <= tol
  np.testing.assert_allclose([float(l) for l in losses], z['E/losses'], atol=10 * tol)


def test_style_argmax_margin_recorded(golden_dir):
  z = _load(golden_dir, 'c2r_fp32')
  assert z['G/pse_margin'].min() > 1e-3   # "bit-exact argmax" is a meaningful check on these inputs
  assert z['G/pse_argmax'].shape == (4,)


def test_state_dict_schema_m8():
  model = O.build_gan(M=8, S=8)
  sd = model.state_dict()
  assert len(sd) == 409
  assert sum(k.startswith('G.') for k in sd) == 391 and sum(k.startswith('D.') for k in sd) == 18
  n_g = sum(p.numel() for p in model.G.parameters() if p.requires_grad)
  n_d = sum(p.numel() for p in model.D.parameters())
  assert (n_g, n_d) == (20151784, 192705)


def test_padding_rule_and_curriculum():
  assert O.default_padding(3, 1) == 1 and O.default_padding(4, 2) == 1 and O.default_padding(4, 1) == 1
  assert O.default_padding((3, 8), 1) == (1, 3) and O.default_padding((3, 3), (1, 1)) == (0, 0)
  c = O.Curriculum(0, 1, 4)
  assert [round(c.step(True), 6) for _ in range(6)] == [0, 0.25, 0.5, 0.75, 1, 1]
  assert c.step(False) == 1.0
