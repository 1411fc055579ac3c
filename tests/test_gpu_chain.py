"""GPU: the chained pose decoder -- decoder.0-3 + logits + softmax mixture in ONE launch (ms_decoder_chain_fwd; reference
src/model/joint_late_cluster_soft_style.py:69-83,106-115,186-194) -- against the same blocks run one by one (same kernels' backward
pass either way), against float64 arithmetic, under HIP-graph replay, repeated launches and load on a second stream."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _build(M, P=104, cin_extra=10, seed=0):
  import mix_stage_amd as A
  torch.manual_seed(seed)
  blocks = [A.ConvNormRelu(256 + cin_extra, 256, type='1d', leaky=True, downsample=False, groups=M)]
  blocks += [A.ConvNormRelu(256, 256, type='1d', leaky=True, downsample=False, groups=M) for _ in range(3)]
  logits = nn.Conv1d(256 * M, P * M, kernel_size=1, stride=1, groups=M)
  g = torch.Generator().manual_seed(seed + 1)
  for m in blocks:
    with torch.no_grad():
      m.norm.weight.copy_(0.5 + torch.rand(m.norm.weight.shape, generator=g))
      m.norm.bias.copy_(torch.randn(m.norm.bias.shape, generator=g) * 0.1)
      m.norm.running_mean.copy_(torch.randn(m.norm.running_mean.shape, generator=g) * 0.1)
      m.norm.running_var.copy_(0.5 + torch.rand(m.norm.running_var.shape, generator=g))
  mods = nn.ModuleList(blocks + [logits]).to(DEV)
  return list(mods[:4]), mods[4]


def _inputs(B, M, cin, seed=0, T=64):
  g = torch.Generator().manual_seed(100 + seed)
  x = torch.randn(B, cin, T, generator=g).to(DEV)
  score = (torch.randn(B, M, T, generator=g) * 2).to(DEV)
  return x, score


def _run(blocks, logits, x, score, P, chain, train=True, grad=True, dout=None):
  """One forward (+ backward) of the segment; returns a dict of everything observable."""
  from mix_stage_amd import ops
  from mix_stage_amd.layers import bare_conv
  for m in blocks:
    m.train(train)
  state = [(m.norm.running_mean.clone(), m.norm.running_var.clone()) for m in blocks]
  params = [p for m in blocks for p in m.parameters()] + list(logits.parameters())
  for p in params:
    p.grad = None
  x = x.clone().requires_grad_(grad)
  score = score.clone().requires_grad_(grad)
  prev = ops.USE_DECODER_CHAIN
  ops.USE_DECODER_CHAIN = chain
  try:
    with torch.set_grad_enabled(grad):
      res = ops.decoder_chain(x, blocks, logits, score, P)
      assert (res is not None) == chain
      if res is None:
        z = blocks[0].forward_broadcast(x)
        for m in blocks[1:]:
          z = m(z)
        z = bare_conv(logits, z, out_f32=True)
        res = ops.softmax_mix(z, score, P)
      out, soft = res
      if grad:
        out.backward(dout)
  finally:
    ops.USE_DECODER_CHAIN = prev
  torch.cuda.synchronize()
  rec = dict(out=out.detach().clone(), soft=soft.detach().clone(),
             running=[(m.norm.running_mean.clone(), m.norm.running_var.clone()) for m in blocks])
  if grad:
    rec['dx'], rec['dscore'] = x.grad.clone(), score.grad.clone()
    rec['grads'] = [p.grad.clone() for p in params]
  for m, (rm, rv) in zip(blocks, state):            # restore: the next run starts from the same running statistics
    with torch.no_grad():
      m.norm.running_mean.copy_(rm)
      m.norm.running_var.copy_(rv)
  return rec


def _close_l2(a, b, tol, what):
  """Relative L2 distance: gradients -- a handful of activations within fp32 rounding of the LeakyReLU kink may take the other
  slope on the two sides, which moves single elements by a few % but not the vector."""
  err = float((a - b).norm()) / max(1e-12, float(b.norm()))
  assert err <= tol, '%s: relative L2 error %.3g' % (what, err)


def _close(a, b, tol, what):
  scale = max(1e-6, float(b.abs().max()))
  err = float((a - b).abs().max()) / scale
  assert err <= tol, '%s: max error %.3g of max |ref| %.3g' % (what, err, scale)


CASES = [('headline', 32, 8, 104, 10), ('m4_b4', 4, 4, 104, 10), ('m25_b8', 8, 25, 104, 10), ('m2_p7_extra16', 3, 2, 7, 16), ('m1', 5, 1, 16, 1)]


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_chain_equals_the_blocks_one_by_one(case):
  """Train mode, forward and backward: the chained launch against conv block after conv block (ms_conv_block_fwd), logits conv
  and ms_softmax_mix_fwd on the same inputs -- the same fp32 products in another summation order."""
  _, B, M, P, extra = case
  blocks, logits = _build(M, P, extra)
  x, score = _inputs(B, M, 256 + extra)
  dout = torch.randn(B, 64, P, generator=torch.Generator().manual_seed(5)).to(DEV)
  a = _run(blocks, logits, x, score, P, True, dout=dout)
  b = _run(blocks, logits, x, score, P, False, dout=dout)
  _close(a['out'], b['out'], 2e-5, 'mixture')
  _close(a['soft'], b['soft'], 1e-6, 'softmax')
  for l, ((rm_a, rv_a), (rm_b, rv_b)) in enumerate(zip(a['running'], b['running'])):
    _close(rm_a, rm_b, 1e-5, 'running mean %d' % l)
    _close(rv_a, rv_b, 1e-5, 'running var %d' % l)
  _close_l2(a['dx'], b['dx'], 1e-3, 'dx')
  _close_l2(a['dscore'], b['dscore'], 1e-3, 'dscore')
  gmax = max(float(g.abs().max()) for g in b['grads'])
  for i, (ga, gb) in enumerate(zip(a['grads'], b['grads'])):
    if i % 4 == 1 and i < 16:
      # the bias in front of a batch-statistics BatchNorm has a gradient of exactly zero: both sides hold rounding noise
      assert float(ga.abs().max()) <= 1e-4 * gmax and float(gb.abs().max()) <= 1e-4 * gmax, i
    else:
      _close_l2(ga, gb, 1e-3, 'gradient of parameter %d' % i)


def test_chain_against_float64():
  """The segment in float64 torch arithmetic (what the reference computes: grouped conv1d, batch-statistics BatchNorm, LeakyReLU,
  1x1 grouped conv, softmax mixture) on the same weights: outputs within the fp32 bar."""
  import torch.nn.functional as F
  B, M, P = 8, 8, 104
  blocks, logits = _build(M, P, 10, seed=3)
  x, score = _inputs(B, M, 266, seed=3)
  a = _run(blocks, logits, x, score, P, True, grad=False)
  h = torch.cat([x.double().cpu()] * M, 1)
  for m in blocks:
    c, n = m.conv, m.norm
    h = F.conv1d(h, c.weight.double().cpu(), c.bias.double().cpu(), padding=1, groups=M)
    mean, var = h.mean((0, 2), keepdim=True), h.var((0, 2), unbiased=False, keepdim=True)
    h = (h - mean) / torch.sqrt(var + n.eps) * n.weight.double().cpu().view(1, -1, 1) + n.bias.double().cpu().view(1, -1, 1)
    h = F.leaky_relu(h, 0.2)
  z = F.conv1d(h, logits.weight.double().cpu(), logits.bias.double().cpu(), groups=M)
  soft = torch.softmax(score.double().cpu().transpose(1, 2), -1)                      # (B, T, M)
  ref = torch.einsum('bgpt,btg->btp', z.view(B, M, P, 64), soft)
  assert float((a['out'].cpu().double() - ref).abs().mean()) <= 2e-6
  _close(a['out'].cpu().double(), ref, 2e-5, 'mixture vs float64')
  _close(a['soft'].cpu().double(), soft, 1e-6, 'softmax vs float64')


def test_chain_eval_mode_equals_blocks():
  """BN_EVAL (the generator's forward inside a D-step, gan.py:106-110): running statistics, no meeting for them."""
  B, M, P = 32, 8, 104
  blocks, logits = _build(M, P, 10, seed=7)
  x, score = _inputs(B, M, 266, seed=7)
  a = _run(blocks, logits, x, score, P, True, train=False, grad=False)
  b = _run(blocks, logits, x, score, P, False, train=False, grad=False)
  _close(a['out'], b['out'], 2e-5, 'mixture (eval)')
  for (rm_a, rv_a), (rm_b, rv_b) in zip(a['running'], b['running']):
    assert torch.equal(rm_a, rm_b) and torch.equal(rv_a, rv_b)                       # untouched


def test_chain_is_bitwise_repeatable_under_load_and_in_a_graph():
  """Three input sets in rotation, 24 launches beside a stream that hammers HBM: every result bit for bit equal to the first launch
  of its set (the meetings hand every workgroup the same complete partials; the monotonic counters need no reset); the same
  launch replayed from a HIP graph gives the same bits; no meeting timed out."""
  from mix_stage_amd import ops16
  B, M, P = 32, 8, 104
  blocks, logits = _build(M, P, 10, seed=11)
  sets = []
  for s in (1, 2, 3):
    x, score = _inputs(B, M, 266, seed=s)
    sets.append((x, score, _run(blocks, logits, x, score, P, True, grad=False)))
  assert not torch.equal(sets[0][2]['out'], sets[1][2]['out'])
  side = torch.cuda.Stream()
  big = torch.randn(64 << 20, device=DEV)
  with torch.cuda.stream(side):
    for _ in range(12):
      big = big * 1.0001 + 0.5
  for rep in range(24):
    x, score, first = sets[rep % 3]
    r = _run(blocks, logits, x, score, P, True, grad=False)
    assert torch.equal(r['out'], first['out']), rep
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(r['running'], first['running'])), rep
  torch.cuda.synchronize()
  # graph replay
  from mix_stage_amd import ops
  x, score, first = sets[0]
  for m in blocks:
    m.train(True)
  state = [(m.norm.running_mean.clone(), m.norm.running_var.clone()) for m in blocks]
  cs = torch.cuda.Stream()
  cs.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(cs), torch.no_grad():
    ops.decoder_chain(x, blocks, logits, score, P)          # warm-up on the capture stream (scratch, counters, weight streams)
  torch.cuda.current_stream().wait_stream(cs)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=cs), torch.no_grad():
    out, soft = ops.decoder_chain(x, blocks, logits, score, P)
  for _ in range(3):
    for m, (rm, rv) in zip(blocks, state):
      m.norm.running_mean.copy_(rm); m.norm.running_var.copy_(rv)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, first['out'])
  assert not ops16.bn_sync_error()


def test_chain_keeps_the_variance_of_a_channel_with_a_huge_mean():
  """bias = 1000 sigma on one channel: per-clip (mean, M2) partials combined by Chan's rule keep its variance."""
  B, M, P = 16, 8, 104
  blocks, logits = _build(M, P, 10, seed=13)
  with torch.no_grad():
    blocks[1].conv.bias[300] = 1000.0
  x, score = _inputs(B, M, 266, seed=13)
  a = _run(blocks, logits, x, score, P, True, grad=False)
  b = _run(blocks, logits, x, score, P, False, grad=False)
  assert torch.isfinite(a['out']).all()
  _close(a['out'], b['out'], 1e-3, 'mixture with a large-mean channel')
  rv_a, rv_b = a['running'][1][1], b['running'][1][1]
  assert abs(float(rv_a[300]) - float(rv_b[300])) <= 2e-2 * abs(float(rv_b[300]))


def test_an_expired_meeting_is_reported_and_poisons_the_output():
  """A counter that is out of step with the launch (set by hand here: one arrival too many on block 0's counter of group 0, what a
  foreign launch on the same words would leave) makes ONE workgroup wait for arrivals that never come: the bounded spin expires
  (~0.3 s), that clip's output is NaN, word 0 of the sync buffer names the meeting, MixStageTrainStep.check_health raises and
  clears; the next launch (counters re-zeroed) is clean."""
  from mix_stage_amd import ops, ops16
  from mix_stage_amd.train_step import MixStageTrainStep
  B, M, P = 8, 4, 104
  blocks, logits = _build(M, P, 10, seed=21)
  x, score = _inputs(B, M, 266, seed=21)
  good = _run(blocks, logits, x, score, P, True, grad=False)
  assert not ops16.bn_sync_error()
  buf = ops16.chain_sync(torch.device(DEV), B, M, 1)
  first = ops.CHAIN_SYNC_FIRST_WORD + int((buf[ops.CHAIN_SYNC_FIRST_WORD:] != 0).nonzero()[0])    # block 0's counter of group 0
  assert int(buf[first]) % B == 0 and int(buf[first]) > 0
  buf[first] += 1
  bad = _run(blocks, logits, x, score, P, True, grad=False)
  assert not torch.isfinite(bad['out']).all()
  assert ops16.bn_sync_error()
  assert any(w[0] != 0 for w in ops16.bn_sync_words())          # which meeting gave up (the peers that wait for the late workgroup at block 1 may expire too and overwrite block 0's code)
  # the raised error word is STICKY: with the counter repaired by hand the next launch meets normally, yet still gives up
  # (NaN) -- an expired launch may leave counters out of step in ways nobody can see, so nothing is trusted until the host
  # has cleared the buffer
  buf[first] -= 1
  still_bad = _run(blocks, logits, x, score, P, True, grad=False)
  assert not torch.isfinite(still_bad['out']).all()
  with pytest.raises(RuntimeError, match='meeting timed out'):
    ops16.check_meetings()
  assert not ops16.bn_sync_error()                               # cleared (all words: the counters start again from zero)
  again = _run(blocks, logits, x, score, P, True, grad=False)
  assert torch.equal(again['out'], good['out'])


def test_channels_that_do_not_invert_keep_y_raw_and_their_gradients():
  """The backward pass of a chained block reads the block's OUTPUT where BatchNorm + LeakyReLU inverts safely; the chain keeps
  y_raw only for the other channels (here: gamma = 1e-6, and beta = 10 gamma).  Gradients must equal the blocks run one by one
  (whose launches keep y_raw everywhere) on those channels as on the rest."""
  B, M, P = 8, 4, 104
  blocks, logits = _build(M, P, 10, seed=31)
  with torch.no_grad():
    blocks[1].norm.weight[7] = 1e-6
    blocks[2].norm.weight[300] = 0.05
    blocks[2].norm.bias[300] = 0.6
    blocks[0].norm.weight[1000] = -0.7                   # negative gamma inverts fine
  x, score = _inputs(B, M, 266, seed=31)
  g = torch.Generator().manual_seed(5)
  dout = torch.randn(B, 64, P, generator=g).to(DEV)
  a = _run(blocks, logits, x, score, P, True, dout=dout)
  b = _run(blocks, logits, x, score, P, False, dout=dout)
  _close(a['out'], b['out'], 2e-5, 'mixture')
  _close_l2(a['dx'], b['dx'], 2e-3, 'dx')                  # (a few activations at the LeakyReLU kink take the other slope)
  gmax = max(float(g.abs().max()) for g in b['grads'])
  for i, (ga, gb) in enumerate(zip(a['grads'], b['grads'])):
    if i % 4 == 1 and i < 16:                              # conv bias in front of a batch-statistics BatchNorm: rounding noise
      assert float(ga.abs().max()) <= 1e-4 * gmax and float(gb.abs().max()) <= 1e-4 * gmax, i
    else:
      _close_l2(ga, gb, 2e-3, 'gradient of parameter %d' % i)
  # the channels that fail the inversion test in particular: gamma / beta gradients of block 1 channel 7 and block 2 channel 300
  for blk, ch in ((1, 7), (2, 300)):
    for k in (2, 3):                                       # parameters per block: conv.weight, conv.bias, norm.weight, norm.bias
      ga, gb = a['grads'][4 * blk + k][ch], b['grads'][4 * blk + k][ch]
      assert abs(float(ga) - float(gb)) <= 2e-3 * max(abs(float(gb)), 1e-3 * gmax), (blk, ch, k, float(ga), float(gb))


def test_shapes_outside_the_chain_fall_back():
  """T != 64, more workgroups than compute units, hooks on a block: decoder_chain declines and the caller runs the blocks."""
  from mix_stage_amd import ops
  blocks, logits = _build(4, 104, 10)
  x = torch.randn(2, 266, 32, device=DEV)
  assert ops.decoder_chain(x, blocks, logits, torch.randn(2, 4, 32, device=DEV), 104) is None
  blocks, logits = _build(25, 104, 10)
  x = torch.randn(32, 266, 64, device=DEV)
  with torch.no_grad():
    assert ops.decoder_chain(x, blocks, logits, torch.randn(32, 25, 64, device=DEV), 104) is None        # 800 workgroups
  blocks, logits = _build(4, 104, 10)
  h = blocks[2].register_forward_hook(lambda m, i, o: None)
  x = torch.randn(4, 266, 64, device=DEV)
  with torch.no_grad():
    assert ops.decoder_chain(x, blocks, logits, torch.randn(4, 4, 64, device=DEV), 104) is None
    h.remove()
    assert ops.decoder_chain(x, blocks, logits, torch.randn(4, 4, 64, device=DEV), 104) is not None


# ------------------------------------------------------------------------------------------------ 16-bit modes
def _run16(blocks, logits, x32, score, P, chain, dt_name, train=True, grad=True, dout=None):
  import mix_stage_amd as A
  from mix_stage_amd import ops, ops16
  from mix_stage_amd.layers import bare_conv
  mods = nn.ModuleList(list(blocks) + [logits])
  A.set_compute_dtype(mods, dt_name)
  dt = ops16.NAME_DT[dt_name]
  for m in blocks:
    m.train(train)
  state = [(m.norm.running_mean.clone(), m.norm.running_var.clone()) for m in blocks]
  params = [p for m in blocks for p in m.parameters()] + list(logits.parameters())
  for p in params:
    p.grad = None
  x32 = x32.clone().requires_grad_(grad)
  score = score.clone().requires_grad_(grad)
  prev = ops.USE_DECODER_CHAIN
  ops.USE_DECODER_CHAIN = chain
  try:
    with torch.set_grad_enabled(grad):
      x = ops16.to_cb8(x32, dt)
      res = ops16.decoder_chain16(x, blocks, logits, score, P)
      assert (res is not None) == chain
      if res is None:
        z = blocks[0].forward_broadcast(x)
        for m in blocks[1:]:
          z = m(z)
        z = bare_conv(logits, z, out_f32=True)
        res = ops.softmax_mix(z, score, P)
      out, soft = res
      if grad:
        out.backward(dout)
  finally:
    ops.USE_DECODER_CHAIN = prev
  torch.cuda.synchronize()
  rec = dict(out=out.detach().clone(), soft=soft.detach().clone(),
             running=[(m.norm.running_mean.clone(), m.norm.running_var.clone()) for m in blocks])
  if grad:
    rec['dx'], rec['dscore'] = x32.grad.clone(), score.grad.clone()
    rec['grads'] = [p.grad.clone() for p in params]
  for m, (rm, rv) in zip(blocks, state):
    with torch.no_grad():
      m.norm.running_mean.copy_(rm)
      m.norm.running_var.copy_(rv)
  return rec


def _segment_float64(blocks, logits, x, score, M, P, round_to):
  """The segment in float64 on operands rounded to the 16-bit type (weights and input), exact arithmetic in between."""
  import torch.nn.functional as F
  r = lambda t: t.detach().to(round_to).double().cpu()
  B = x.shape[0]
  h = torch.cat([r(x)] * M, 1)
  for m in blocks:
    c, n = m.conv, m.norm
    h = F.conv1d(h, r(c.weight), c.bias.double().cpu(), padding=1, groups=M)
    mean, var = h.mean((0, 2), keepdim=True), h.var((0, 2), unbiased=False, keepdim=True)
    h = (h - mean) / torch.sqrt(var + n.eps) * n.weight.double().cpu().view(1, -1, 1) + n.bias.double().cpu().view(1, -1, 1)
    h = F.leaky_relu(h, 0.2)
  z = F.conv1d(h, r(logits.weight), logits.bias.double().cpu(), groups=M)
  soft = torch.softmax(score.double().cpu().transpose(1, 2), -1)
  return torch.einsum('bgpt,btg->btp', z.view(B, M, P, 64), soft)


@pytest.mark.parametrize('dt_name', ['bf16', 'fp16'])
@pytest.mark.parametrize('case', [('headline', 32, 8, 104, 10), ('m4_b4', 4, 4, 104, 10), ('m3_p16_extra16', 5, 3, 16, 16)], ids=lambda c: c[0])
def test_chain16_against_blocks_and_float64(case, dt_name):
  """16-bit chain: the same roundings as the blocks one by one (16-bit operands, fp32 accumulators normalised from registers, one
  rounding of every block output) -- both sides within the same distance of exact arithmetic on the rounded operands, and the
  gradients (computed by the blocks' own backward pass from what the chain stored) agree."""
  _, B, M, P, extra = case
  tdt = torch.bfloat16 if dt_name == 'bf16' else torch.float16
  blocks, logits = _build(M, P, extra, seed=21)
  x, score = _inputs(B, M, 256 + extra, seed=21)
  dout = torch.randn(B, 64, P, generator=torch.Generator().manual_seed(6)).to(DEV)
  a = _run16(blocks, logits, x, score, P, True, dt_name, dout=dout)
  b = _run16(blocks, logits, x, score, P, False, dt_name, dout=dout)
  ref = _segment_float64(blocks, logits, x, score, M, P, tdt)
  ea = float((a['out'].cpu().double() - ref).abs().mean())
  eb = float((b['out'].cpu().double() - ref).abs().mean())
  scale = float(ref.abs().mean())
  print('%s %s: chain %.3e, blocks %.3e of mean |out| %.3e' % (case[0], dt_name, ea, eb, scale))
  assert torch.isfinite(a['out']).all()
  assert ea <= 1.5 * eb + 1e-4 * scale and ea <= (3e-2 if dt_name == 'bf16' else 5e-3) * scale
  _close(a['soft'], b['soft'], 1e-6, 'softmax')
  for l, ((rm_a, rv_a), (rm_b, rv_b)) in enumerate(zip(a['running'], b['running'])):
    _close(rm_a, rm_b, 2e-2, 'running mean %d' % l)
    _close(rv_a, rv_b, 2e-2, 'running var %d' % l)
  tol = 6e-2 if dt_name == 'bf16' else 1.5e-2
  _close_l2(a['dx'], b['dx'], tol, 'dx')
  _close_l2(a['dscore'], b['dscore'], tol, 'dscore')
  gmax = max(float(g.abs().max()) for g in b['grads'])
  for i, (ga, gb) in enumerate(zip(a['grads'], b['grads'])):
    if i % 4 == 1 and i < 16:
      assert float(ga.abs().max()) <= 2e-2 * gmax, i            # bias in front of BatchNorm: zero up to rounding
    else:
      _close_l2(ga, gb, tol, 'gradient of parameter %d' % i)


def test_chain16_eval_and_repeatability():
  from mix_stage_amd import ops16
  B, M, P = 32, 8, 104
  blocks, logits = _build(M, P, 10, seed=23)
  x, score = _inputs(B, M, 266, seed=23)
  a = _run16(blocks, logits, x, score, P, True, 'bf16', train=False, grad=False)
  b = _run16(blocks, logits, x, score, P, False, 'bf16', train=False, grad=False)
  ref_scale = float(b['out'].abs().mean())
  assert float((a['out'] - b['out']).abs().mean()) <= 2e-2 * ref_scale
  first = _run16(blocks, logits, x, score, P, True, 'bf16', grad=False)
  side = torch.cuda.Stream()
  big = torch.randn(32 << 20, device=DEV)
  with torch.cuda.stream(side):
    for _ in range(8):
      big = big * 1.0001 + 0.5
  for rep in range(12):
    r = _run16(blocks, logits, x, score, P, True, 'bf16', grad=False)
    assert torch.equal(r['out'], first['out']), rep
  torch.cuda.synchronize()
  assert not ops16.bn_sync_error()
