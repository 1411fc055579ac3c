"""GPU parity of the whole path (generator, discriminator, GAN D-step / G-step) against the CPU oracle and the
golden vectors captured from the reference.  Bars (BASELINE.json north_star): mean |pose_hip - pose_ref| <= 1e-4
on the fp32 path, style-id argmax bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import mixstage_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def build_hip_gan(M, S, T=64, P=104):
  import mix_stage_amd as A
  G = A.JointLateClusterSoftStyle4_G(time_steps=T, out_feats=P, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  D = A.Speech2Gesture_D(in_channels=P)
  model = A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)
  model.load_state_dict(O.deterministic_state(model.state_dict()))
  model.G.thresh.value, model.G.thresh.iters = 1, 10 ** 9
  return model.to(DEV)


def _step(model, batch, kind, dev):
  audio, pose, labels, style = [t.to(dev) for t in batch]
  model.train(); model.zero_grad()
  model.D_prob = 1.1 if kind == 'D' else -1.0
  fake, losses, _ = model([audio, labels], pose, **O.model_kwargs(style))
  total = 0
  for l in losses:
    total = total + l
  total.backward()
  return fake, losses


@pytest.mark.parametrize('M,S,B', [(4, 4, 4), (1, 2, 4), (8, 8, 2)])
@pytest.mark.parametrize('kind', ['G', 'D'])
def test_gan_step_matches_oracle(M, S, B, kind):
  batch = O.synthetic_batch(B, M=M, S=S)
  ref = O.build_gan(M=M, S=S)
  hip = build_hip_gan(M, S)
  f_ref, l_ref = _step(ref, batch, kind, 'cpu')
  f_hip, l_hip = _step(hip, batch, kind, DEV)
  l1 = (f_hip.detach().cpu() - f_ref.detach()).abs().mean().item()
  assert l1 <= 1e-4, 'pose L1 %g' % l1
  for a, b in zip(l_hip, l_ref):
    assert abs(float(a) - float(b)) <= 1e-4, (float(a), float(b))
  assert (hip.G.labels_cap_soft.cpu() - ref.G.labels_cap_soft.detach()).abs().max().item() <= 1e-4
  assert hip.G_flag == ref.G_flag
  bad = []
  for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
    if q.grad is None:
      if p.grad is not None and p.grad.abs().max().item() != 0 and not (kind == 'G' and n.startswith('D.')):
        bad.append((n, 'unexpected grad'))
      continue
    if p.grad is None:
      if not (kind == 'G' and n.startswith('D.')):     # D weight grads are skipped in the G-step by design
        bad.append((n, 'missing grad'))
      continue
    scale = q.grad.abs().max().item()
    err = (p.grad.cpu() - q.grad).abs().max().item()
    is_pre_bn_bias = n.endswith('conv.bias')           # true gradient is 0 (BN removes the bias)
    tol = 2e-3 * scale + 1e-6 if not is_pre_bn_bias else 1e-5
    if err > tol:
      bad.append((n, err, scale))
  assert not bad, bad[:10]
  for (k, a), (_, b) in zip(hip.state_dict().items(), ref.state_dict().items()):
    if 'running_' in k:
      assert (a.cpu() - b).abs().max().item() <= 1e-4 * (1 + b.abs().max().item()), k
    if k.endswith('num_batches_tracked'):
      assert int(a) == int(b), k


@pytest.mark.parametrize('name', ['c1_fp32', 'c2r_fp32', 'c3r_fp32'])
def test_golden_vectors_from_reference(golden_dir, name):
  z = np.load(os.path.join(golden_dir, name + '.npz'))
  B, T, M, S = [int(v) for v in z['meta']]
  batch = [torch.from_numpy(z[k]) for k in ('audio', 'pose', 'labels', 'style')]
  for kind in ('G', 'D'):
    hip = build_hip_gan(M, S)
    seen = {}

    def hook(m, i, o):
      seen.setdefault('pse', o.detach())

    h = hip.G.pose_style_encoder.register_forward_hook(hook)
    fake, losses = _step(hip, batch, kind, DEV)
    h.remove()
    k = kind + '/'
    assert np.abs(fake.detach().cpu().numpy() - z[k + 'pose']).mean() <= 1e-4
    np.testing.assert_allclose([float(l) for l in losses], z[k + 'losses'], atol=1e-4)
    np.testing.assert_allclose(hip.G.labels_cap_soft.cpu().numpy(), z[k + 'labels_cap_soft'], atol=1e-4)
    if kind == 'G':
      assert np.array_equal(seen['pse'].argmax(-1).cpu().numpy(), z[k + 'pse_argmax'])    # bit-exact style ids
      np.testing.assert_allclose(seen['pse'].cpu().numpy(), z[k + 'pse_score'], atol=1e-4)
  hip = build_hip_gan(M, S).eval()
  audio, pose, labels, style = [t.to(DEV) for t in batch]
  with torch.no_grad():
    fake, losses, _ = hip([audio, labels], pose, **O.model_kwargs(style))
    kw = O.model_kwargs(style); kw['sample_flag'] = 1
    fake_s, _, _ = hip([audio, labels], pose, **kw)
  assert np.abs(fake.cpu().numpy() - z['E/pose']).mean() <= 1e-4
  assert np.abs(fake_s.cpu().numpy() - z['S/pose']).mean() <= 1e-4


def test_repeatability_bitwise():
  batch = O.synthetic_batch(2, M=2, S=2)
  outs = []
  for _ in range(2):
    hip = build_hip_gan(2, 2)
    fake, _ = _step(hip, batch, 'G', DEV)
    outs.append((fake.detach().clone(), hip.G.decoder[0].conv.weight.grad.clone()))
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_double_model_fails_loudly():
  hip = build_hip_gan(2, 2).double()
  audio, pose, labels, style = [t.to(DEV) for t in O.synthetic_batch(2, M=2, S=2, dtype=torch.float64)]
  hip.eval()
  with pytest.raises(TypeError):
    hip([audio, labels], pose, **O.model_kwargs(style))
