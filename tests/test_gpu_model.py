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


def _record_block_outputs(model, is_block):
  """forward hooks on every conv block: name -> list of outputs (to detect LeakyReLU kink flips)."""
  rec, handles = {}, []
  for name, m in model.named_modules():
    if is_block(m):
      def hook(mod, i, o, name=name):
        rec.setdefault(name, []).append(o.detach())
      handles.append(m.register_forward_hook(hook))
  return rec, handles


FLIP_LOG = []


def _count_kink_flips(rec_hip, rec_ref):
  flips = 0
  for name, outs in rec_ref.items():
    outs_h = rec_hip[name]
    if 2 * len(outs_h) == len(outs) and outs_h[0].shape[0] == 2 * outs[0].shape[0]:
      # the D-step's paired discriminator pass: one call on the batch [fake; real] where the reference makes two calls
      outs_h = [h for o in outs_h for h in (o[:o.shape[0] // 2], o[o.shape[0] // 2:])]
    for a, b in zip(outs_h, outs):
      n = int(((a.cpu() > 0) != (b > 0)).sum())
      if n:
        bad = (a.cpu() > 0) != (b > 0)
        FLIP_LOG.append((name, n, tuple(a.shape), float(a.cpu()[bad].abs().max()), float(b[bad].abs().max())))
      flips += n
  return flips


def _compare_step(M, S, B, kind, F_, seed, strict):
  """HIP vs the fp64 oracle.  Returns the number of activations that sit within fp32 rounding of a LeakyReLU
  kink and took the other slope (the derivative is discontinuous there, so gradients are only compared tightly on
  flip-free draws)."""
  import mix_stage_amd as A
  batch = O.synthetic_batch(B, M=M, S=S, F_=F_, seed=seed)
  ref = O.build_gan(M=M, S=S, dtype=torch.float64)
  hip = build_hip_gan(M, S)
  rec_r, h_r = _record_block_outputs(ref, lambda m: isinstance(m, O.ConvNormRelu))
  rec_h, h_h = _record_block_outputs(hip, lambda m: isinstance(m, A.ConvNormRelu))
  batch64 = [t.double() if t.is_floating_point() else t for t in batch]
  f_ref, l_ref = _step(ref, batch64, kind, 'cpu')
  f_hip, l_hip = _step(hip, batch, kind, DEV)
  for h in h_r + h_h:
    h.remove()
  flips = _count_kink_flips(rec_h, rec_r)
  l1 = (f_hip.detach().cpu().double() - f_ref.detach()).abs().mean().item()
  assert l1 <= 1e-4, 'pose L1 %g' % l1
  for a, b in zip(l_hip, l_ref):
    assert abs(float(a) - float(b)) <= 1e-4, (float(a), float(b))
  assert (hip.G.labels_cap_soft.cpu().double() - ref.G.labels_cap_soft.detach()).abs().max().item() <= 1e-4
  assert hip.G_flag == ref.G_flag
  if strict and flips:
    return flips
  # flip-free: every element within 2e-3 of the parameter's max |grad|.  With flips (a handful of activations out of
  # ~10^6-10^7 took the other LeakyReLU slope) single elements move by a few %, so the bar is on the relative L2
  # error of each gradient tensor (3 %) plus a loose per-element cap (15 %).
  bad = []
  for (n, p), (_, q) in zip(hip.named_parameters(), ref.named_parameters()):
    skipped_by_design = kind == 'G' and n.startswith('D.')     # D weight grads are not computed in the G-step
    if q.grad is None:
      if p.grad is not None and p.grad.abs().max().item() != 0 and not skipped_by_design:
        bad.append((n, 'unexpected grad'))
      continue
    if p.grad is None:
      if not skipped_by_design:
        bad.append((n, 'missing grad'))
      continue
    scale = q.grad.abs().max().item()
    diff = p.grad.cpu().double() - q.grad
    err = diff.abs().max().item()
    if n.endswith('conv.bias'):      # conv bias in front of BN: the true gradient is 0, both sides hold rounding noise
      if err > 5e-5:   # fp32 rounding noise of a sum that is exactly 0 in real arithmetic (other gradients: 1e-3..1)
        bad.append((n, err, scale))
      continue
    if not flips:
      if err > 2e-3 * scale + 1e-7:
        bad.append((n, err, scale))
    else:
      l2 = diff.norm().item() / (q.grad.norm().item() + 1e-30)
      if err > 0.15 * scale + 1e-7 or l2 > 3e-2:
        bad.append((n, err, scale, l2))
  assert not bad, 'flips=%d %s' % (flips, bad[:8])
  for (k, a), (_, b) in zip(hip.state_dict().items(), ref.state_dict().items()):
    if 'running_' in k:
      assert (a.cpu().double() - b).abs().max().item() <= 1e-4 * (1 + b.abs().max().item()), k
    if k.endswith('num_batches_tracked'):
      assert int(a) == int(b), k
  return flips


@pytest.mark.parametrize('M,S,B', [(4, 4, 2), (1, 2, 2)])
@pytest.mark.parametrize('kind', ['G', 'D'])
def test_gan_step_gradients_strict(M, S, B, kind):
  """All parameter gradients within 2e-3 of the fp64 oracle on a draw without kink flips (16 mel bins keep the
  activation count, hence the flip probability, small)."""
  for attempt in range(5):
    if _compare_step(M, S, B, kind, F_=16, seed=100 + attempt, strict=True) == 0:
      return
  # every draw had an activation within ~1e-6 of a kink that took the other slope: compare at the loose bar
  _compare_step(M, S, B, kind, F_=16, seed=100, strict=False)


@pytest.mark.parametrize('M,S,B', [(4, 4, 4), (8, 8, 2)])
@pytest.mark.parametrize('kind', ['G', 'D'])
def test_gan_step_matches_oracle(M, S, B, kind):
  """Full-size mel axis: outputs/losses at the 1e-4 bar; gradients at 2e-3 (flip-free) or 3 % relative L2 (a few activations
  flipped slope at a kink -- expected with ~10^7 activations in fp32)."""
  _compare_step(M, S, B, kind, F_=128, seed=1234, strict=False)


@pytest.mark.parametrize('precision', ['fp32', 'bf16x6'])
@pytest.mark.parametrize('kind', ['G', 'D'])
def test_headline_size_step_matches_oracle(kind, precision):
  """The size bench.py times (B=32, T=64, 128-mel, M=S=8: 1024-workgroup launches, 4 workgroups per CU, full-grid XCD
  remap) against the fp64 oracle: pose, losses, softmax, every parameter gradient, BN running statistics -- in both
  arithmetic modes of the fp32 path."""
  from mix_stage_amd import _lib
  prev = _lib.lib().ms_set_precision(1 if precision == 'bf16x6' else 0)
  try:
    _compare_step(8, 8, 32, kind, F_=128, seed=1234, strict=False)
  finally:
    _lib.lib().ms_set_precision(prev)


@pytest.mark.parametrize('name', ['c1_fp32', 'c2r_fp32', 'c3r_fp32'])
def test_golden_gradients_from_reference(golden_dir, name):
  """Every parameter gradient of the HIP path against the vectors the REFERENCE produced (make_golden.py: L2 norm and
  a 16-element strided sample of each gradient, recorded after clip_grad_norm_ rescaled them in place)."""
  z = np.load(os.path.join(golden_dir, name + '.npz'))
  B, T, M, S = [int(v) for v in z['meta']]
  batch = [torch.from_numpy(z[k]) for k in ('audio', 'pose', 'labels', 'style')]
  from mix_stage_amd import ops, _lib
  for kind, forms in (('G', 'default'), ('D', 'default'), ('D', 'block by block')):
    # 'block by block': the decoder blocks one by one and the per-layer conv kernels instead of the chained launch and the
    # clip-resident blocks -- the generator's summation order closest to plain per-layer arithmetic.  Both forms are held to the
    # same 5 % bar: the fixtures' data seeds are screened so that no pre-activation of D sits within 2e-5 of its LeakyReLU kink
    # (tests/golden/screen_seed.py; with the unscreened seed of rounds 1-5 one sat at 6.7e-6, a 1e-7 perturbation of the fake pose
    # moved D's gradients by 0.5-2 % in L2, and the default forms needed a 20 % bar).
    old_chain, old_clip = ops.USE_DECODER_CHAIN, None
    if forms != 'default':
      ops.USE_DECODER_CHAIN = False
      old_clip = _lib.lib().ms_debug_set_clip32(0)
    try:
      hip = build_hip_gan(M, S)
      _step(hip, batch, kind, DEV)
    finally:
      ops.USE_DECODER_CHAIN = old_chain
      if old_clip is not None:
        _lib.lib().ms_debug_set_clip32(old_clip)
    k = kind + '/'
    mod = hip.G if kind == 'G' else hip.D
    grads = {kind + '.' + n: p.grad.detach().double().reshape(-1).cpu() for n, p in mod.named_parameters()
             if p.grad is not None}
    total = float(torch.sqrt(sum((g * g).sum() for g in grads.values())))
    ref_total = float(z[k + 'total_grad_norm'])
    assert abs(total - ref_total) <= 1e-3 * ref_total, (total, ref_total)
    coef = min(1.0, 1.0 / (ref_total + 1e-6))          # the reference probed its gradients after the clip
    worst = 0.0
    for n, g in grads.items():
      if k + 'gnorm/' + n not in z.files:
        assert float(g.abs().max()) == 0.0, n            # the reference left this parameter without a gradient
        continue
      gn, gs = float(z[k + 'gnorm/' + n]), z[k + 'gsamp/' + n]
      stride = max(1, g.numel() // 16)
      mine = (g[::stride][:16] * coef).numpy()
      if n.endswith('conv.bias') and gn < 1e-4 * ref_total * coef:
        continue                                         # conv bias in front of BN: exactly 0 in real arithmetic
      scale = gn / np.sqrt(g.numel()) + 1e-12            # rms of the gradient: the per-element yardstick
      assert abs(float(g.norm()) * coef - gn) <= 2e-3 * gn + 1e-9, (n, float(g.norm()) * coef, gn)
      worst = max(worst, float(np.abs(mine - gs).max() / scale))
    # sampled elements within 5 % of the gradient's rms, production forms included
    assert worst <= 5e-2, (kind, forms, worst)


def test_d_step_gradients_do_not_depend_on_the_generator_forms_beyond_the_kink(golden_dir):
  """The D-step's discriminator gradients with the generator's default forms (chained decoder, clip-resident blocks) against the
  block-by-block forms, with the SAME fake pose fed to D in both: identical arithmetic in D then, so the gradients must agree to
  fp32 rounding (D's own kernels do not care which generator form produced their input)."""
  from mix_stage_amd import ops, _lib
  z = np.load(os.path.join(golden_dir, 'c2r_fp32.npz'))
  B, T, M, S = [int(v) for v in z['meta']]
  batch = [torch.from_numpy(z[k]) for k in ('audio', 'pose', 'labels', 'style')]
  hip = build_hip_gan(M, S)
  fake, _ = _step(hip, batch, 'D', DEV)
  fake = fake.detach()
  g_default = {n: p.grad.detach().clone() for n, p in hip.D.named_parameters() if p.grad is not None}
  # the same discriminator pass on that very fake pose, generator forms switched: only D's own launches matter now
  old_chain = ops.USE_DECODER_CHAIN
  old_clip = _lib.lib().ms_debug_set_clip32(0)
  ops.USE_DECODER_CHAIN = False
  try:
    hip2 = build_hip_gan(M, S)
    real_forward = hip2.G.forward
    hip2.G.forward = lambda *a, **kw: (fake, real_forward(*a, **kw)[1])
    _step(hip2, batch, 'D', DEV)
  finally:
    ops.USE_DECODER_CHAIN = old_chain
    _lib.lib().ms_debug_set_clip32(old_clip)
  for n, p in hip2.D.named_parameters():
    if p.grad is None:
      continue
    a, b = g_default[n].double(), p.grad.detach().double()
    if n.endswith('conv.bias') and 'conv1' not in n and 'logits' not in n:
      continue                                         # conv bias in front of BN: rounding noise on both sides
    assert (a - b).norm().item() <= 2e-4 * (b.norm().item() + 1e-12), (n, (a - b).norm().item(), b.norm().item())


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


def test_curriculum_pose_branch_and_thresh_ramp():
  """Row a11: while the curriculum threshold is below the host draw, the generator is fed PoseEncoder(y) instead of
  the audio encoder (JL:127-129); thresh ramps by 1/1000 per training call (layers.py:677-696)."""
  M = S = 2
  batch = O.synthetic_batch(3, M=M, S=S)
  ref = O.build_gan(M=M, S=S)
  hip = build_hip_gan(M, S)
  for m in (ref, hip):
    m.G.thresh.value, m.G.thresh.iters = 0.0, 0          # fresh curriculum: the first draws pick the pose branch
  for step in range(2):
    outs = []
    for m, dev in ((ref, 'cpu'), (hip, DEV)):
      audio, pose, labels, style = [t.to(dev) for t in batch]
      m.train(); m.zero_grad()
      m.D_prob = -1.0
      torch.manual_seed(21 + step)
      fake, losses, _ = m([audio, labels], pose, **O.model_kwargs(style))
      sum(l for l in losses if l.requires_grad).backward()
      outs.append((fake.detach().cpu(), [float(l) for l in losses]))
    assert (outs[0][0] - outs[1][0]).abs().mean().item() <= 1e-4
    np.testing.assert_allclose(outs[1][1], outs[0][1], atol=2e-4)
    assert hip.G.thresh.value == ref.G.thresh.value == (step + 1) / 1000
    # the pose branch trains pose_encoder and leaves audio_encoder without gradients
    for m in (ref, hip):
      assert m.G.pose_encoder.conv[0].conv.weight.grad is not None
      assert m.G.audio_encoder.conv[0].conv.weight.grad is None
    g_ref = ref.G.pose_encoder.conv[2].conv.weight.grad
    g_hip = hip.G.pose_encoder.conv[2].conv.weight.grad.cpu()
    assert ((g_hip - g_ref).norm() / g_ref.norm()).item() < 2e-2


def test_long_context_full_speaker_set():
  """BASELINE configs[3] geometry: M = S = 25 (full PATS speaker set), T = 256 (time axis tiled inside the kernels,
  BN statistics over the full (B,T)); reduced batch so the CPU oracle finishes in seconds."""
  M = S = 25
  T = 256
  batch = O.synthetic_batch(2, T=T, M=M, S=S)
  ref = O.build_gan(M=M, S=S, T=T)
  import mix_stage_amd as A
  G = A.JointLateClusterSoftStyle4_G(time_steps=T, out_feats=104, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  D = A.Speech2Gesture_D(in_channels=104)
  hip = A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)
  hip.load_state_dict(O.deterministic_state(hip.state_dict()))
  hip.G.thresh.value, hip.G.thresh.iters = 1, 10 ** 9
  hip = hip.to(DEV)
  for kind in ('G', 'D'):
    for m, dev in ((ref, 'cpu'), (hip, DEV)):
      audio, pose, labels, style = [t.to(dev) for t in batch]
      m.train(); m.zero_grad()
      m.D_prob = 1.1 if kind == 'D' else -1.0
      fake, losses, _ = m([audio, labels], pose, **O.model_kwargs(style, T))
      sum(l for l in losses if l.requires_grad).backward()
      if dev == 'cpu':
        f_ref, l_ref = fake.detach(), [float(l) for l in losses]
      else:
        f_hip, l_hip = fake.detach().cpu(), [float(l) for l in losses]
    assert f_hip.shape == (2, T, 104)
    assert (f_hip - f_ref).abs().mean().item() <= 1e-4
    np.testing.assert_allclose(l_hip, l_ref, atol=2e-4)
    probe = (lambda m: m.G.decoder[1].conv.weight) if kind == 'G' else (lambda m: m.D.conv3.conv.weight)
    g_ref, g_hip = probe(ref).grad, probe(hip).grad.cpu()
    assert ((g_hip - g_ref).norm() / g_ref.norm()).item() < 2e-2


def test_inference_style_transfer_batch():
  """BASELINE configs[4] path: eval mode, sample_flag=1 (style ids given, PoseStyleEncoder bypassed), large batch."""
  B, M, S = 96, 8, 8
  audio, pose, labels, style = O.synthetic_batch(B, M=M, S=S)
  style = (style + 3) % S                     # transfer to another speaker's style (trainer.py:1367-1386)
  ref = O.build_gan(M=M, S=S).eval()
  hip = build_hip_gan(M, S).eval()
  kw = O.model_kwargs(style); kw['sample_flag'] = 1
  with torch.no_grad():
    f_ref, l_ref, _ = ref([audio, labels], pose, **kw)
    kw_h = dict(kw); kw_h['style'] = style.to(DEV)
    f_hip, l_hip, _ = hip([audio.to(DEV), labels.to(DEV)], pose.to(DEV), **kw_h)
  assert (f_hip.cpu() - f_ref).abs().mean().item() <= 1e-4
  assert abs(float(l_hip[0]) - float(l_ref[0])) <= 1e-4
  assert hip.G.labels_cap_soft.shape == (B, 64, M)


@pytest.mark.parametrize('use_graphs', [False, True])
def test_sampling_driver_long_sequence_all_styles(use_graphs):
  """The reference's sampling loop core (trainer.py:779-786, 1367-1386): windows of one interval concatenated into one
  long sequence, one eval forward per target style -- vs the oracle doing the same on the CPU."""
  from mix_stage_amd.sample import StyleTransferSampler
  M = S = 4
  n = 6                                               # 6 windows -> one sequence of 384 frames
  audio, pose, labels, style = O.synthetic_batch(n, M=M, S=S)
  style = torch.full_like(style, 2)                   # one interval = one speaker
  ref = O.build_gan(M=M, S=S).eval()
  hip = build_hip_gan(M, S)
  sampler = StyleTransferSampler(hip, num_styles=S, use_graphs=use_graphs)
  for rep in range(2):                                # second call replays the captured graph
    torch.manual_seed(5)
    got = sampler.sample_interval(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV))
    assert [g[0] for g in got] == [None, '2_3', '2_0', '2_1']
    torch.manual_seed(5)
    for (name, y_hip, l_hip), shift in zip(got, range(S)):
      kw = O.model_kwargs(((style + shift) % S).reshape(1, -1), T=n * 64)
      kw.update(sample_flag=1, desc='test', description='test')
      with torch.no_grad():
        y_ref, l_ref, _ = ref([audio.reshape(1, -1, 128), labels.reshape(1, -1)], pose.reshape(1, -1, 104), **kw)
      assert y_hip.shape == (1, n * 64, 104)
      assert (y_hip.cpu() - y_ref).abs().mean().item() <= 1e-4, name
      assert abs(float(l_hip[0]) - float(l_ref[0])) <= 1e-4


def test_repeatability_bitwise():
  batch = O.synthetic_batch(2, M=2, S=2)
  outs = []
  for _ in range(2):
    hip = build_hip_gan(2, 2)
    fake, _ = _step(hip, batch, 'G', DEV)
    outs.append((fake.detach().clone(), hip.G.decoder[0].conv.weight.grad.clone()))
  assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_double_model_runs_like_the_reference_trainer_casts_it():
  """trainer.py:138 casts the model with .to(device).double() and dataUtils.py:547 feeds float64 batches.  The mirrored
  modules accept exactly that: parameters, buffers and gradients stay float64 tensors (state_dict compatibility), the
  kernels compute on fp32 shadows -- results within the fp32 bar of the float64 oracle."""
  M = S = 2
  batch64 = O.synthetic_batch(3, M=M, S=S, dtype=torch.float64)
  ref = O.build_gan(M=M, S=S, dtype=torch.float64)
  hip = build_hip_gan(M, S).to(DEV).double()
  assert all(p.dtype == torch.float64 for p in hip.parameters())
  for kind in ('G', 'D'):
    f_ref, l_ref = _step(ref, batch64, kind, 'cpu')
    f_hip, l_hip = _step(hip, batch64, kind, DEV)
    assert f_hip.dtype == torch.float64
    assert (f_hip.detach().cpu() - f_ref.detach()).abs().mean().item() <= 1e-4
    for a, b in zip(l_hip, l_ref):
      assert abs(float(a) - float(b)) <= 1e-4
    mod_h, mod_r = (hip.G, ref.G) if kind == 'G' else (hip.D, ref.D)
    for (n, p), (_, q) in zip(mod_h.named_parameters(), mod_r.named_parameters()):
      if q.grad is None or n.endswith('conv.bias'):
        continue
      assert p.grad is not None and p.grad.dtype == torch.float64, n
      assert ((p.grad.cpu() - q.grad).norm() / (q.grad.norm() + 1e-30)).item() <= 3e-2, n
  for (k, a), (_, b) in zip(hip.state_dict().items(), ref.state_dict().items()):
    if 'running_' in k:
      assert a.dtype == torch.float64 and (a.cpu() - b).abs().max().item() <= 1e-4 * (1 + b.abs().max().item()), k


@pytest.mark.parametrize('kind', ['G', 'D'])
def test_mse_criterion_the_constructor_default(kind):
  """GAN(criterion='MSELoss') -- gan.py:21,40: the constructor default -- on the fused squared-error kernels."""
  import mix_stage_amd as A
  M = S = 2
  batch = O.synthetic_batch(3, M=M, S=S)
  ref = O.build_gan(M=M, S=S, dtype=torch.float64)
  ref.criterion = torch.nn.MSELoss(reduction='none')
  G = A.JointLateClusterSoftStyle4_G(time_steps=64, out_feats=104, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  hip = A.GAN(G, A.Speech2Gesture_D(in_channels=104), input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)
  assert hip.criterion_name == 'MSELoss'
  hip.load_state_dict(O.deterministic_state(hip.state_dict()))
  hip.G.thresh.value, hip.G.thresh.iters = 1, 10 ** 9
  hip = hip.to(DEV)
  batch64 = [t.double() if t.is_floating_point() else t for t in batch]
  f_ref, l_ref = _step(ref, batch64, kind, 'cpu')
  f_hip, l_hip = _step(hip, batch, kind, DEV)
  assert (f_hip.detach().cpu().double() - f_ref.detach()).abs().mean().item() <= 1e-4
  for a, b in zip(l_hip, l_ref):
    assert abs(float(a) - float(b)) <= 1e-4
  probe = (lambda m: m.G.logits.weight) if kind == 'G' else (lambda m: m.D.conv3.conv.weight)
  g_ref, g_hip = probe(ref).grad, probe(hip).grad.cpu().double()
  assert ((g_hip - g_ref).norm() / g_ref.norm()).item() <= 2e-2


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_reference_format_weight_file_on_the_device(tmp_path, dtype):
  """N4 (README.md:124-141, trainer.py:138,148): a pickled `*_weights.p` state_dict of GAN(G, D) -- fp32, or fp64 as a
  `.double()`-cast reference model writes it, DataParallel prefix and wrapper key included -- loaded with
  checkpoint.load_weights into the HIP modules reproduces what the oracle computes from the SAME file: train-mode G-step and
  D-step outputs and losses, eval-mode sampling output."""
  import pickle
  from mix_stage_amd.checkpoint import load_weights
  M = S = 3
  torch.manual_seed(11)
  src = O.build_gan(M=M, S=S)
  with torch.no_grad():                                     # not the deterministic fill the other tests use: trained-looking weights
    for n, p in src.named_parameters():
      p.add_(0.02 * torch.randn_like(p))
    for n, b in src.named_buffers():
      if b.is_floating_point():
        b.add_(0.05 * torch.rand_like(b))
  if dtype == 'float64':
    src = src.double()
  path = tmp_path / 'exp_42_weights.p'
  with open(path, 'wb') as f:
    pickle.dump({'model': {'module.' + k: v.clone() for k, v in src.state_dict().items()}}, f)
  ref = O.build_gan(M=M, S=S, dtype=getattr(torch, dtype))
  ref.load_state_dict(src.state_dict())
  hip = build_hip_gan(M, S)
  if dtype == 'float64':
    hip = hip.double()                                      # what trainer.py:138 does to the model
  res = load_weights(hip, str(path))
  assert not res.missing_keys and not res.unexpected_keys
  batch = O.synthetic_batch(4, M=M, S=S, seed=3)
  if dtype == 'float64':
    batch = [t.double() if t.is_floating_point() else t for t in batch]
  for kind in ('G', 'D'):
    f_ref, l_ref = _step(ref, batch, kind, 'cpu')
    f_hip, l_hip = _step(hip, batch, kind, DEV)
    assert (f_hip.detach().cpu().double() - f_ref.detach().double()).abs().mean().item() <= 1e-4
    np.testing.assert_allclose([float(l) for l in l_hip], [float(l) for l in l_ref], atol=2e-4)
  audio, pose, labels, style = batch
  kw = O.model_kwargs((style + 1) % S); kw['sample_flag'] = 1
  ref.eval(); hip.eval()
  with torch.no_grad():
    y_ref, _, _ = ref([audio, labels], pose, **kw)
    kw_d = dict(kw); kw_d['style'] = kw['style'].to(DEV)
    y_hip, _, _ = hip([audio.to(DEV), labels.to(DEV)], pose.to(DEV), **kw_d)
  assert (y_hip.cpu().double() - y_ref.double()).abs().mean().item() <= 1e-4


@pytest.mark.parametrize('B,T', [(32, 64), (4, 64), (6, 32)])
def test_paired_discriminator_pass_equals_the_two_passes(B, T):
  """Speech2Gesture_D.forward_pair (one batch of 2B clips, BatchNorm statistics per half: MS_DT_STAT_PAIR) against the two passes of
  gan.py:120,126 one after the other: scores, every gradient, the running statistics after their two sequential updates and the
  tracked batch counts.  Tolerances are fp32 summation order (the batch of 2B is tiled and split differently), not statistics."""
  import copy
  import mix_stage_amd as A
  from mix_stage_amd import ops
  torch.manual_seed(5)
  D1 = A.Speech2Gesture_D(in_channels=104).to(DEV).train()
  with torch.no_grad():
    for m in D1.modules():
      if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
  D2 = copy.deepcopy(D1)
  fake = torch.randn(B, 104, T, device=DEV)
  real = torch.randn(B, 104, T, device=DEV) * 1.7 + 0.4          # different statistics in the two halves
  assert D2.pair_supported(torch.empty(2 * B, 104, T, device='meta')), 'the paired form is not offered at this shape'
  s_f = D1.forward_channel_major(fake)[0]
  s_r = D1.forward_channel_major(real)[0]
  (ops.l1_mean(s_f, target=0.0, scale=0.7) + ops.l1_mean(s_r, target=1.0)).backward()
  p_f, p_r = D2.forward_pair(torch.cat([fake, real], dim=0))
  (ops.l1_mean(p_f, target=0.0, scale=0.7) + ops.l1_mean(p_r, target=1.0)).backward()
  torch.cuda.synchronize()
  assert torch.allclose(p_f, s_f, rtol=1e-4, atol=1e-5) and torch.allclose(p_r, s_r, rtol=1e-4, atol=1e-5)
  for (n, a), (_, b) in zip(D1.named_parameters(), D2.named_parameters()):
    if n.endswith('conv.bias'):
      continue                           # a conv bias in front of BatchNorm: its true gradient is zero, both sides hold rounding noise
    err = (a.grad - b.grad).norm().item()
    assert err <= 2e-4 * (a.grad.norm().item() + 1e-8), (n, err, a.grad.norm().item())
  sd1, sd2 = D1.state_dict(), D2.state_dict()
  for k in sd1:
    if 'running_' in k:
      assert torch.allclose(sd1[k], sd2[k], rtol=1e-5, atol=1e-6), k
    if 'num_batches_tracked' in k:
      assert int(sd1[k]) == int(sd2[k]) == 2, (k, int(sd1[k]), int(sd2[k]))
  # one half changed: the OTHER half's scores do not move (its statistics are its own)
  D3 = copy.deepcopy(D2)
  q_f, q_r = D3.forward_pair(torch.cat([fake, real * 3.0 - 1.0], dim=0))
  d_f, d_r = D2.forward_pair(torch.cat([fake, real], dim=0))
  # (D2 and D3 differ in running statistics only, which train-mode outputs do not read)
  assert torch.equal(q_f, d_f) and not torch.allclose(q_r, d_r)


def test_d_step_with_paired_passes_equals_the_step_with_two_passes(golden_dir):
  """GAN.forward's D-step through the paired discriminator pass (the default) against the same step with the two passes one after the
  other (pair_D_passes = False): losses, D's gradients, D's running statistics."""
  z = np.load(os.path.join(golden_dir, 'c2r_fp32.npz'))
  B, T, M, S = [int(v) for v in z['meta']]
  batch = [torch.from_numpy(z[k]) for k in ('audio', 'pose', 'labels', 'style')]
  out = []
  for pair in (True, False):
    hip = build_hip_gan(M, S)
    hip.pair_D_passes = pair
    torch.manual_seed(1)
    _, losses = _step(hip, batch, 'D', DEV)
    out.append((hip, [float(l.detach()) for l in losses]))
  (h1, l1), (h2, l2) = out
  assert np.allclose(l1, l2, rtol=1e-5, atol=1e-6), (l1, l2)
  for (n, a), (_, b) in zip(h1.D.named_parameters(), h2.D.named_parameters()):
    if n.endswith('conv.bias'):
      continue
    err = (a.grad - b.grad).norm().item()
    assert err <= 2e-4 * (b.grad.norm().item() + 1e-8), (n, err, b.grad.norm().item())
  for (n, a), (_, b) in zip(h1.D.named_buffers(), h2.D.named_buffers()):
    if 'running_' in n:
      assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), n


def test_paired_pass_steps_aside_for_hooked_modules():
  """A forward hook on one of D's blocks expects the reference's two calls per D-step on batches of B: the paired pass declines."""
  import mix_stage_amd as A
  D = A.Speech2Gesture_D(in_channels=104).to(DEV).train()
  probe = torch.empty(64, 104, 64, device='meta')
  assert D.pair_supported(probe)
  calls = []
  h = D.conv3.register_forward_hook(lambda m, i, o: calls.append(o.shape[0]))
  assert not D.pair_supported(probe)
  h.remove()
  assert D.pair_supported(probe)
