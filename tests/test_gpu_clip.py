"""GPU: the clip-resident 1-D conv blocks (csrc/clip32.hip) -- Conv1d k3 s1 p1 (plain input or nearest_up2(a) + r) and k4 s2 p1,
[+ BatchNorm1d(train / eval) + LeakyReLU], whole block in one launch with the batch statistics met inside the launch -- against the
fp64 oracle block (reference layers.py:32-78) at the depths of the UNet (T = 64 ... 2), the classifier, the style encoder and the
discriminator, forward and backward (data gradient by the same kernel with transposed weights; weight gradient by the existing
kernels from what the block stored)."""
import zlib

import pytest
import torch

from test_gpu_kernels import _conv_block_case, _mk_block, rel_err, DEV
from oracle import mixstage_oracle as O

pytestmark = pytest.mark.gpu

# (name, type, cin, cout, kernel, stride, groups, input spatial, in_mode)
CASES = [
    ('k3_t64', '1d', 256, 256, None, None, 1, (64,), 'plain'),
    ('k3_t32', '1d', 256, 256, None, None, 1, (32,), 'plain'),
    ('k3_t8', '1d', 256, 256, None, None, 1, (8,), 'plain'),
    ('k3_t2', '1d', 256, 256, None, None, 1, (2,), 'plain'),
    ('k3_cin266', '1d', 266, 256, None, None, 1, (64,), 'plain'),
    ('k3_104_64', '1d', 104, 64, None, None, 1, (64,), 'plain'),
    ('up2_t64', '1d', 256, 256, None, None, 1, (64,), 'up2'),
    ('up2_t16', '1d', 256, 256, None, None, 1, (16,), 'up2'),
    ('up2_t4', '1d', 256, 256, None, None, 1, (4,), 'up2'),
    ('s2_t64', '1d', 256, 256, 4, 2, 1, (64,), 'plain'),
    ('s2_t16', '1d', 256, 256, 4, 2, 1, (16,), 'plain'),
    ('s2_t4', '1d', 256, 256, 4, 2, 1, (4,), 'plain'),
    ('s2_64_128', '1d', 64, 128, 4, 2, 1, (32,), 'plain'),
    ('s2_128_256', '1d', 128, 256, 4, 2, 1, (8,), 'plain'),
    ('s2_t2', '1d', 256, 256, 4, 2, 1, (2,), 'plain'),             # PoseStyleEncoder's last steps: one output frame per clip
    ('s2_256_8_t2', '1d', 256, 8, 4, 2, 1, (2,), 'plain'),         # ... its 8-way style head (rows of the channel tile masked)
    ('s2_104_64', '1d', 104, 64, 4, 2, 1, (64,), 'plain'),         # D.conv1's geometry (104 pose features in)
    ('k3_256_40', '1d', 256, 40, None, None, 1, (64,), 'plain'),   # output channels that do not fill the second channel tile
]


def _labels(fn):
  from mix_stage_amd import ops
  ops.timing_enable(True)
  try:
    fn()
    torch.cuda.synchronize()
    return [r['label'] for r in ops.timing_report()]
  finally:
    ops.timing_enable(False)


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_clip_block_train_fwd_bwd_vs_oracle(case):
  B = 32
  labels = _labels(lambda: _conv_block_case(case, B, 7))
  assert any('conv_fwd_clip' in l for l in labels), labels
  assert not any('bn_finalize' in l or 'splitk_fwd_epilogue' in l for l in labels), labels
  # the data gradient runs on the clip kernel as well: k3 s1 by the transposed, tap-reversed stream; k4 s2 as one 2-tap conv per
  # output parity
  assert any('conv_dgrad_clip' in l for l in labels), labels
  assert not any('splitk_dgrad_epilogue' in l for l in labels), labels
  for attempt in range(4):
    if _conv_block_case(case, B, zlib.crc32(case[0].encode()) % 1000 + attempt):
      return
  raise AssertionError('no draw without a sign flip at a LeakyReLU kink')


@pytest.mark.parametrize('case', [CASES[0], CASES[4], CASES[9], CASES[12]], ids=lambda c: c[0])
def test_clip_block_eval_mode(case):
  import mix_stage_amd as A
  name, typ, cin, cout, k, s, g, sp, in_mode = case
  gen = torch.Generator().manual_seed(5)
  ref = _mk_block(O, case).double().eval()
  hip = _mk_block(A, case).to(DEV).eval()
  x = torch.randn(32, cin * g, *sp, generator=gen)
  with torch.no_grad():
    labels = _labels(lambda: hip(x.to(DEV)))
    y = hip(x.to(DEV))
    y_ref = ref(x.double())
  assert any('conv_fwd_clip' in l for l in labels), labels
  assert rel_err(y, y_ref) < 2e-5
  assert rel_err(hip.norm.running_mean, ref.norm.running_mean) == 0.0


def test_clip_block_is_bitwise_repeatable_and_replays_in_a_graph():
  """20 launches of a BN_TRAIN block on two input sets in rotation beside a stream that hammers HBM: bit-identical outputs per
  set (the in-launch meeting hands every workgroup the same complete partials; monotonic counters), the same from a HIP graph,
  and no meeting timed out."""
  import mix_stage_amd as A
  from mix_stage_amd import ops16
  case = CASES[0]
  hip = _mk_block(A, case).to(DEV).train()
  xs = [torch.randn(32, 256, 64, generator=torch.Generator().manual_seed(s)).to(DEV) for s in (1, 2)]
  state = (hip.norm.running_mean.clone(), hip.norm.running_var.clone())

  def run(x):
    with torch.no_grad():
      hip.norm.running_mean.copy_(state[0]); hip.norm.running_var.copy_(state[1])
      return hip(x).clone(), hip.norm.running_mean.clone()
  firsts = [run(x) for x in xs]
  assert not torch.equal(firsts[0][0], firsts[1][0])
  side = torch.cuda.Stream()
  big = torch.randn(32 << 20, device=DEV)
  with torch.cuda.stream(side):
    for _ in range(8):
      big = big * 1.0001 + 0.5
  for rep in range(20):
    y, rm = run(xs[rep % 2])
    assert torch.equal(y, firsts[rep % 2][0]) and torch.equal(rm, firsts[rep % 2][1]), rep
  torch.cuda.synchronize()
  cs = torch.cuda.Stream()
  cs.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(cs), torch.no_grad():
    hip(xs[0])
  torch.cuda.current_stream().wait_stream(cs)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=cs), torch.no_grad():
    yg = hip(xs[0])
  for _ in range(3):
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(yg, firsts[0][0])
  assert not ops16.bn_sync_error()


@pytest.mark.parametrize('kind', ['classifier', 'style_encoder'])
def test_stack_backward_fuses_the_producers_batchnorm_into_the_data_gradient(kind):
  """Inside a 1-D stack (layers.py: ClusterClassify.conv 446-467, PoseStyleEncoder.conv 246-289) a block's output feeds the next
  block only, so the backward pass lets the NEXT block's data-gradient launch carry this block's BatchNorm + LeakyReLU backward
  (ms_bwd_options.prev_*: the workgroups of a channel tile meet inside the launch for the batch sums).  Gradients of every parameter
  and of the input against the fp64 oracle stack, the launch list (one bn_bwd launch left: the last block's), and against the
  unfused backward pass of the same modules."""
  import mix_stage_amd as A
  from mix_stage_amd import ops
  from mix_stage_amd.train_step import FlatAdam
  B = 8
  gen = torch.Generator().manual_seed(3)
  if kind == 'classifier':
    ref = O.ClusterClassify(num_clusters=8, input_channels=266).double().train()
    hip = A.ClusterClassify(num_clusters=8, input_channels=266)
    x = torch.randn(B, 266, 64, generator=gen)
    call = lambda m, t: m(t)
    n_blocks = 6
  else:
    B = 32            # (the deepest levels hold B * T = 64 frames: one workgroup's worth only from 32 clips on)
    ref = O.PoseStyleEncoder(input_channels=104, num_speakers=8).double().train()
    hip = A.PoseStyleEncoder(input_channels=104, num_speakers=8)
    x = torch.randn(B, 64, 104, generator=gen)
    call = lambda m, t: m(t)
    n_blocks = 7
  sd = O.deterministic_state(ref.state_dict())
  ref.load_state_dict(sd)
  hip.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in sd.items()})
  hip = hip.to(DEV).train()
  opt = FlatAdam(hip.parameters())
  x64 = x.double().requires_grad_()
  y_ref = call(ref, x64)
  gy = torch.randn(y_ref.shape, generator=gen)
  y_ref.backward(gy.double())

  def run(chain):
    opt.zero_grad()
    xh = x.to(DEV).requires_grad_()
    if not chain:
      mods = list(hip.conv)
      saved = [m.forward for m in mods]
      for m in mods:      # the same modules with the chain hint dropped
        m.forward = (lambda mm: (lambda t, **kw: type(mm).forward(mm, t, **{k: v for k, v in kw.items() if k != '_ms_chain'})))(m)
    try:
      ops.timing_enable(True)
      try:
        call(hip, xh).backward(gy.to(DEV))
        torch.cuda.synchronize()
        lab = {r['label'].split('|')[-1]: r['count'] for r in ops.timing_report()}
      finally:
        ops.timing_enable(False)
    finally:
      if not chain:
        for m, f in zip(mods, saved):
          del m.forward
    return xh.grad.clone(), opt.flat_g.clone(), lab

  dx_f, g_f, lab_f = run(True)
  dx_u, g_u, lab_u = run(False)
  n_bn_f = sum(c for l, c in lab_f.items() if 'bn_bwd' in l)
  n_bn_u = sum(c for l, c in lab_u.items() if 'bn_bwd' in l)
  assert n_bn_u == n_blocks and n_bn_f == 1, (lab_f, lab_u)
  assert sum(c for l, c in lab_f.items() if 'dgrad_clip' in l and 'ep7' in l) == n_blocks - 1
  assert rel_err(dx_f, x64.grad) < 1e-4 and rel_err(dx_u, x64.grad) < 1e-4
  for (n, p), o in zip(hip.named_parameters(), opt.offsets):
    gref = dict(ref.named_parameters())[n].grad
    gf = g_f[o:o + p.numel()].view_as(p)
    gu = g_u[o:o + p.numel()].view_as(p)
    if 'conv.bias' in n and not n.startswith('logits'):
      # a conv bias in front of BatchNorm: the true gradient is 0, every side holds rounding noise
      assert gf.abs().max().item() <= 1e-4 * max(1.0, gref.abs().max().item()), n
      continue
    assert rel_err(gf, gref) < 1e-4, (n, rel_err(gf, gref))
    assert rel_err(gf, gu.double()) < 1e-5, (n, rel_err(gf, gu.double()))


@pytest.mark.parametrize('how', ['forward_hook_into_loss', 'tensor_hook', 'backward_hook'])
def test_a_watched_intermediate_output_keeps_the_unfused_backward(how):
  """The fused form hands the producing block dy_raw in place of the gradient of its OUTPUT, which is only right while that
  output feeds the next block alone and nobody looks at its gradient (advisor, round 5).  A forward hook that puts an interior
  block's output into the loss (a second consumer), a tensor hook on it, or a module backward hook must each see and produce the
  gradients of the unfused backward pass: hooked blocks are left out of the fusion (layers._hooked, ops.conv_block)."""
  import mix_stage_amd as A
  from mix_stage_amd import ops
  from mix_stage_amd.train_step import FlatAdam
  B = 8
  gen = torch.Generator().manual_seed(5)
  ref = O.ClusterClassify(num_clusters=8, input_channels=266).double().train()
  hip = A.ClusterClassify(num_clusters=8, input_channels=266)
  sd = O.deterministic_state(ref.state_dict())
  ref.load_state_dict(sd)
  hip.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in sd.items()})
  hip = hip.to(DEV).train()
  opt = FlatAdam(hip.parameters())
  x = torch.randn(B, 266, 64, generator=gen)
  gy = torch.randn(B, 8, 64, generator=gen)
  taps, seen = {}, {}

  def attach(model, store):
    blk = model.conv[2]
    if how == 'forward_hook_into_loss':
      return [blk.register_forward_hook(lambda m, i, o: store.__setitem__('y', o))]
    if how == 'tensor_hook':
      return [blk.register_forward_hook(lambda m, i, o: o.register_hook(lambda g: store.__setitem__('g', g.detach().clone())) and None)]
    return [blk.register_full_backward_hook(lambda m, gi, go: store.__setitem__('g', go[0].detach().clone()))]

  def loss_of(y, store, g):
    l = (y * g).sum()
    if how == 'forward_hook_into_loss':
      l = l + 0.37 * (store['y'].double() if y.dtype == torch.float64 else store['y']).pow(2).sum()
    return l

  hs = attach(ref, taps)
  x64 = x.double().requires_grad_()
  loss_of(ref(x64), taps, gy.double()).backward()
  for h in hs:
    h.remove()
  hs = attach(hip, seen)
  opt.zero_grad()
  xh = x.to(DEV).requires_grad_()
  ops.timing_enable(True)
  try:
    loss_of(hip(xh), seen, gy.to(DEV)).backward()
    torch.cuda.synchronize()
    lab = {r['label'].split('|')[-1]: r['count'] for r in ops.timing_report()}
  finally:
    ops.timing_enable(False)
    for h in hs:
      h.remove()
  # blocks 2 (hooked) and 3 (its consumer) stay out of the fusion: block 2's and block 1's BatchNorm backward run on their own
  n_bn = sum(c for l, c in lab.items() if 'bn_bwd' in l)
  assert n_bn >= 3, lab
  assert rel_err(xh.grad, x64.grad) < 1e-4, rel_err(xh.grad, x64.grad)
  if 'g' in taps:
    assert rel_err(seen['g'], taps['g']) < 1e-4, rel_err(seen['g'], taps['g'])
  for (n, p), o in zip(hip.named_parameters(), opt.offsets):
    gref = dict(ref.named_parameters())[n].grad
    gf = opt.flat_g[o:o + p.numel()].view_as(p)
    if 'conv.bias' in n and not n.startswith('logits'):
      continue
    assert rel_err(gf, gref) < 1e-4, (n, rel_err(gf, gref))


def test_fusion_is_dropped_when_meetings_are_switched_off_after_the_forward_pass():
  """ops16.set_in_launch_meetings(False) between forward and backward (block_sync then returns None): the backward pass falls back
  to the blocks' own BatchNorm backward instead of failing half-way (advisor, round 5)."""
  import mix_stage_amd as A
  from mix_stage_amd import ops16
  from mix_stage_amd.train_step import FlatAdam
  gen = torch.Generator().manual_seed(3)       # (the draw of the stack test above: no activation within fp32 rounding of a LeakyReLU kink)
  ref = O.ClusterClassify(num_clusters=8, input_channels=266).double().train()
  hip = A.ClusterClassify(num_clusters=8, input_channels=266)
  sd = O.deterministic_state(ref.state_dict())
  ref.load_state_dict(sd)
  hip.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in sd.items()})
  hip = hip.to(DEV).train()
  opt = FlatAdam(hip.parameters())
  x = torch.randn(8, 266, 64, generator=gen)
  gy = torch.randn(8, 8, 64, generator=gen)
  x64 = x.double().requires_grad_()
  ref(x64).backward(gy.double())
  def run(switch_off):
    opt.zero_grad()
    xh = x.to(DEV).requires_grad_()
    y = hip(xh)
    old = ops16.in_launch_meetings()
    if switch_off:
      ops16.set_in_launch_meetings(False)
    try:
      y.backward(gy.to(DEV))
      torch.cuda.synchronize()
    finally:
      ops16.set_in_launch_meetings(old)
    return xh.grad.clone(), opt.flat_g.clone()

  dx_f, g_f = run(False)
  dx_o, g_o = run(True)
  assert rel_err(dx_f, x64.grad) < 1e-3 and rel_err(dx_o, x64.grad) < 1e-3
  assert rel_err(dx_o, dx_f) < 1e-4, rel_err(dx_o, dx_f)
  assert rel_err(g_o, g_f) < 1e-4, rel_err(g_o, g_f)


@pytest.mark.parametrize('hooked', [False, True])
def test_unet_residual_gradients_meet_inside_the_down_blocks_data_gradient(hooked):
  """UNet1D (layers.py:80-157): every down-path output feeds the next down block and, as the residual, the up path.  The up block's
  residual gradient is handed to the down block's data-gradient launch (ms_bwd_options.dx_accum, ops.ResidualLink) instead of being
  added by an accumulation launch per level.  Input and parameter gradients against the fp64 oracle and against the same modules
  with the links off; a hooked block keeps autograd's own accumulation."""
  import mix_stage_amd as A
  from mix_stage_amd import ops
  from mix_stage_amd.train_step import FlatAdam
  B = 32
  gen = torch.Generator().manual_seed(11)
  ref = O.UNet1D(256, 256).double().train()
  hip = A.UNet1D(256, 256)
  sd = O.deterministic_state(ref.state_dict())
  ref.load_state_dict(sd)
  hip.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in sd.items()})
  hip = hip.to(DEV).train()
  opt = FlatAdam(hip.parameters())
  x = torch.randn(B, 256, 64, generator=gen)
  gy = torch.randn(B, 256, 64, generator=gen)
  x64 = x.double().requires_grad_()
  ref(x64).backward(gy.double())
  seen = []
  h = hip.conv1[2].register_forward_hook(lambda m, i, o: seen.append(1)) if hooked else None

  def run(links):
    old = ops.enable_chain_fusion(links)
    try:
      opt.zero_grad()
      xh = x.to(DEV).requires_grad_()
      n0 = ops._link_stats['in_launch']
      hip(xh).backward(gy.to(DEV))
      torch.cuda.synchronize()
      return xh.grad.clone(), opt.flat_g.clone(), ops._link_stats['in_launch'] - n0
    finally:
      ops.enable_chain_fusion(old)

  try:
    dx_l, g_l, n_l = run(True)
    dx_u, g_u, n_u = run(False)
  finally:
    if h is not None:
      h.remove()
  # five levels; a hook on conv1[2] takes out the links in which it is producer (level 3) or consumer (level 2)
  assert n_u == 0 and n_l == (3 if hooked else 5), (n_l, n_u)
  assert rel_err(dx_l, x64.grad) < 1e-3 and rel_err(dx_u, x64.grad) < 1e-3, (rel_err(dx_l, x64.grad), rel_err(dx_u, x64.grad))
  assert rel_err(dx_l, dx_u) < 2e-5, rel_err(dx_l, dx_u)
  assert rel_err(g_l, g_u) < 2e-5, rel_err(g_l, g_u)
