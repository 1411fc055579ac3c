"""GPU: the trainer-step contract (zero_grad -> forward -> sum of losses -> backward -> clip_grad_norm_(.,1) ->
Adam(1e-4) on G or D) against the CPU oracle, eager vs HIP-graph replay, and the golden post-step vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import mixstage_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hip(M, S):
  from test_gpu_model import build_hip_gan
  return build_hip_gan(M, S)


def test_first_step_matches_golden_post_adam_state(golden_dir):
  """After one clipped Adam step from the deterministic weights: loss scalars, global grad norm and parameter
  checksums captured from the reference (tests/golden/make_golden.py)."""
  from mix_stage_amd.train_step import MixStageTrainStep
  z = np.load(os.path.join(golden_dir, 'c2r_fp32.npz'))
  B, T, M, S = [int(v) for v in z['meta']]
  batch = [torch.from_numpy(z[k]).to(DEV) for k in ('audio', 'labels', 'pose', 'style')]
  for kind in ('G', 'D'):
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=False)
    assert ts.step(*batch, kind=kind) == kind
    np.testing.assert_allclose([float(l) for l in ts.losses], z[kind + '/losses'], atol=1e-4)
    opt = ts.optim_G if kind == 'G' else ts.optim_D
    np.testing.assert_allclose(float(opt.norm), z[kind + '/total_grad_norm'], rtol=1e-3)
    mod = model.G if kind == 'G' else model.D
    worst = 0.0
    for n, p in mod.named_parameters():
      ref_sum = float(z['%s/psum/%s.%s' % (kind, kind, n)])
      # first Adam step moves every element by at most lr = 1e-4 (sign-like update)
      worst = max(worst, abs(float(p.detach().double().sum()) - ref_sum) / max(1, p.numel()))
    assert worst <= 2e-5, worst


def test_three_steps_vs_oracle_and_graph_equals_eager():
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 4
  batches = [O.synthetic_batch(4, M=M, S=S, seed=50 + i) for i in range(3)]
  kinds = ['G', 'D', 'G']
  ref = O.build_gan(M=M, S=S)
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  ref_losses = []
  for (audio, pose, labels, style), k in zip(batches, kinds):
    _, l, _ = O.oracle_train_step(ref, og, od, audio, pose, labels, style, k)
    ref_losses.append(l)
  results = {}
  for use_graphs in (False, True):
    torch.manual_seed(99)
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=use_graphs)
    got = []
    # run the sequence twice with graphs so the second pass is pure replay
    for rep in range(2 if use_graphs else 1):
      if rep == 1:
        model.load_state_dict(O.deterministic_state(model.state_dict()))
        for o in (ts.optim_G, ts.optim_D):
          o.reset_state()
        got = []
      for (audio, pose, labels, style), k in zip(batches, kinds):
        ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind=k)
        got.append([float(l) for l in ts.losses])
    results[use_graphs] = (got, {k: v.clone() for k, v in model.state_dict().items()}, ts.optim_G.step_count,
                           ts.optim_D.step_count)
  eager, graph = results[False], results[True]
  assert eager[2:] == (2, 1) and graph[2:] == (2, 1)
  # HIP eager vs oracle
  # step 1 sees identical weights; later steps see weights that differ by Adam's sign-like first updates
  # (an element whose gradient is ~0 moves by +-lr in a direction decided by rounding noise)
  for i, (a, b) in enumerate(zip(eager[0], ref_losses)):
    np.testing.assert_allclose(a, b, atol=2e-4 if i == 0 else 3e-3)
  ref_sd = ref.state_dict()
  for k, v in eager[1].items():
    if v.is_floating_point():
      d = (v.cpu() - ref_sd[k]).abs()
      if 'running_' in k:     # batch statistics of slightly different weights: relative bar
        assert d.max().item() <= 2e-3 * (1 + ref_sd[k].abs().max().item()), (k, d.max().item())
      else:                   # parameters: <= 2 Adam steps * 2 * lr
        assert d.max().item() <= 4.5e-4, (k, d.max().item())
        assert d.mean().item() <= 1e-4, (k, d.mean().item())
    else:
      assert int(v) == int(ref_sd[k]), k
  # graph replay is bit-identical to eager
  assert graph[0] == eager[0]
  for k, v in eager[1].items():
    assert torch.equal(v, graph[1][k]), k


@pytest.mark.parametrize('use_graphs', [False, True])
def test_adam_per_parameter_step_origin_under_the_curriculum(use_graphs):
  """torch.optim.Adam counts steps per parameter from its first gradient and (torch 1.5 zero_grad semantics) keeps
  updating a parameter with zero gradients once it has had one: pose_encoder is trained in pose-branch steps only,
  audio_encoder/unet-input in audio-branch steps only.  Sequence: G(pose) G(audio) D G(pose) G(audio)."""
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 2
  batches = [O.synthetic_batch(3, M=M, S=S, seed=70 + i) for i in range(5)]
  POSE, AUDIO = (0.0, 0), (1.0, 10 ** 9)      # (thresh.value, thresh.iters): fresh curriculum / finished curriculum
  plan = [('G', POSE), ('G', AUDIO), ('D', AUDIO), ('G', POSE), ('G', AUDIO)]
  ref = O.build_gan(M=M, S=S)
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  hip = _hip(M, S)
  ts = MixStageTrainStep(hip, use_graphs=use_graphs)
  for (audio, pose, labels, style), (kind, th) in zip(batches, plan):
    for m in (ref, hip):
      m.G.thresh.value, m.G.thresh.iters = th
    torch.manual_seed(3)
    O.oracle_train_step(ref, og, od, audio, pose, labels, style, kind)
    torch.manual_seed(3)
    ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind=kind)
  # first-gradient steps recorded per parameter: pose_encoder at step 1, audio_encoder at step 2 (G optimizer steps)
  index = {id(p): i for i, p in enumerate(ts.optim_G.params)}        # (the flat buffer is laid out used-first, not in module order)
  first = {n: ts.optim_G.host_first[index[id(p)]] for n, p in hip.G.named_parameters() if p.requires_grad}
  assert first['pose_encoder.conv.0.conv.weight'] == 1 and first['audio_encoder.conv.0.conv.weight'] == 2
  assert first['text_encoder.conv.0.conv.weight'] == -1 and first['unet.conv1.0.conv.weight'] == 1
  ref_sd = ref.state_dict()
  for k, v in hip.state_dict().items():
    if not v.is_floating_point() or 'running_' in k:
      continue
    d = (v.cpu() - ref_sd[k]).abs()
    # <= 4 Adam steps of <= lr each, direction of near-zero gradients decided by rounding noise
    assert d.max().item() <= 8.5e-4, (k, d.max().item())
    assert d.mean().item() <= 1.5e-4, (k, d.mean().item())
  # a never-used parameter is bit-for-bit untouched
  assert torch.equal(hip.state_dict()['G.text_encoder.conv.0.conv.weight'].cpu(), ref_sd['G.text_encoder.conv.0.conv.weight'])


def test_wgrad_side_stream_overlap_is_bit_identical():
  """ms_conv_block_bwd_overlap: weight gradients on a side stream give the same bits as the serial path."""
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 2
  batch = [t.to(DEV) for t in O.synthetic_batch(3, M=M, S=S)]
  audio, pose, labels, style = batch
  out = []
  from mix_stage_amd import ops
  old = ops.enable_chain_fusion(False)       # (the side-stream launches have no fused producer-BatchNorm form: same arithmetic on both sides)
  for overlap in (False, True):
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=False, overlap_wgrad=overlap)
    for k in ('G', 'D', 'G'):
      ts.step(audio, labels, pose, style, kind=k)
    torch.cuda.synchronize()
    out.append({k: v.clone() for k, v in model.state_dict().items()})
  ops.enable_chain_fusion(old)
  for k, v in out[0].items():
    assert torch.equal(v, out[1][k]), k


def test_reference_coin_flip_sequence_and_rng_parity():
  """kind=None: the step kind follows gan.py:105's host draw and consumes the host generator exactly like the
  reference (2 draws per step), eager and under graph replay."""
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 2
  audio, pose, labels, style = O.synthetic_batch(2, M=M, S=S)
  ref = O.build_gan(M=M, S=S)
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4); od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  torch.manual_seed(4321)
  ref_kinds = []
  for _ in range(6):
    ref.train(); ref.zero_grad()
    ref([audio, labels], pose, **O.model_kwargs(style))
    ref_kinds.append('G' if ref.G_flag else 'D')
  tail_ref = torch.rand(1).item()
  for use_graphs in (False, True):
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=use_graphs)
    torch.manual_seed(4321)
    kinds = [ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV)) for _ in range(6)]
    assert kinds == ref_kinds and torch.rand(1).item() == tail_ref
    assert len(set(kinds)) == 2          # the seed exercises both step kinds


def test_prepared_dgrad_weights_and_deferred_wgrad_reductions_change_nothing():
  """Deferred weight-gradient reductions (one ms_wgrad_reduce_multi launch at the end of backward) and: the data-gradient weights are built once per optimizer update (ms_dgrad_weights_prepare) instead of per backward
  call.  They must track (a) the HIP Adam updates inside captured steps and (b) edits torch makes behind the optimizer's
  back (load_state_dict), and the feature must not change a single bit of the parameters."""
  from mix_stage_amd import ops
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 4
  batches = [O.synthetic_batch(4, M=M, S=S, seed=70 + i) for i in range(6)]
  kinds = ['G', 'D', 'G', 'G', 'D', 'G']

  def run(prepared, use_graphs):
    torch.manual_seed(5)
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=use_graphs)
    ops.enable_prepared_weights(prepared)
    ops.enable_deferred_wgrad(prepared)
    # (the planner hint that travels with the queued launches -- longer workgroups, fewer pixel splits -- changes the ORDER of the
    # weight-gradient sums, not the machinery under test: pinned on for both runs)
    ops.lib().ms_set_wgrad_batched(1, 0)
    snap = None
    for i, ((audio, pose, labels, style), k) in enumerate(zip(batches, kinds)):
      if i == 3:      # perturb every parameter the way a checkpoint load does (in-place copy_ into the views)
        sd = {n: v * 1.01 if v.dtype.is_floating_point else v for n, v in model.state_dict().items()}
        model.load_state_dict(sd)
      ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind=k)
    torch.cuda.synchronize()
    snap = (ts.optim_G.flat_p.clone(), ts.optim_D.flat_p.clone(), len(ops._prepared['entries']), len(ops._deferred['bufs']))
    ops.enable_prepared_weights(False)
    ops.enable_deferred_wgrad(False)
    return snap

  base = run(False, False)
  assert base[2] == 0 and base[3] == 0
  for use_graphs in (False, True):
    got = run(True, use_graphs)
    assert got[2] > 0 and got[3] > 0, 'no block used prepared weights / deferred reductions'
    assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), 'prepared weights changed the result'


@pytest.mark.parametrize('precision', [0, 1], ids=['fp32', 'bf16x6'])
def test_prepared_weights_created_after_a_graph_was_captured(precision):
  """A step kind captured first cannot rebuild prepared buffers that only appear later (D's data-gradient weights for the
  G-step's batch shape, G's forward planes for blocks only the D-step runs): they are refreshed eagerly after each replay.
  D-first order, graph replay vs eager, bit for bit, in both arithmetic modes."""
  from mix_stage_amd import _lib
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 4
  batches = [O.synthetic_batch(4, M=M, S=S, seed=90 + i) for i in range(6)]
  kinds = ['D', 'G', 'D', 'G', 'G', 'D']
  old = _lib.lib().ms_set_precision(precision)
  try:
    out = {}
    for use_graphs in (False, True):
      torch.manual_seed(11)
      model = _hip(M, S)
      ts = MixStageTrainStep(model, use_graphs=use_graphs)
      for (audio, pose, labels, style), k in zip(batches, kinds):
        ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind=k)
      torch.cuda.synchronize()
      out[use_graphs] = (ts.optim_G.flat_p.clone(), ts.optim_D.flat_p.clone())
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
  finally:
    _lib.lib().ms_set_precision(old)


@pytest.mark.parametrize('use_graphs', [False, True])
def test_lin_style_path_trains_the_style_embedding(use_graphs):
  """argmax=0 (joint_late_cluster_soft_style.py:160-166): the style embedding is reached through EmbLin's plain matmul, whose
  gradient torch autograd accumulates into the flat-buffer slot behind the kernels' back.  The segmented Adam must still see
  it: style_emb.emb.weight moves like torch.optim.Adam moves it in the oracle."""
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 2
  ref = O.build_gan(M=M, S=S)
  hip = _hip(M, S)
  ref.G.argmax = hip.G.argmax = 0
  og = torch.optim.Adam(ref.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(ref.D.parameters(), lr=1e-4)
  ts = MixStageTrainStep(hip, use_graphs=use_graphs)
  w0 = hip.G.style_emb.emb.weight.detach().clone()
  for i in range(3):
    audio, pose, labels, style = O.synthetic_batch(3, M=M, S=S, seed=80 + i)
    torch.manual_seed(9)
    O.oracle_train_step(ref, og, od, audio, pose, labels, style, 'G')
    torch.manual_seed(9)
    ts.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind='G')
  w_hip, w_ref = hip.G.style_emb.emb.weight.detach().cpu(), ref.G.style_emb.emb.weight.detach()
  assert (w_hip - w0.cpu()).abs().max().item() >= 1e-4          # it was updated at all (3 Adam steps of ~lr each)
  assert (w_hip - w_ref).abs().max().item() <= 1.5e-4, (w_hip - w_ref).abs().max().item()
  # a parameter that never receives a gradient keeps torch's "skipped" semantics: untouched bit for bit
  assert torch.equal(hip.G.smoothen.conv.weight.detach().cpu(), ref.G.smoothen.conv.weight.detach())


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
def test_ramping_lambda_schedule_under_graphs_equals_eager(precision):
  """gan.py:30-33,103: the reference constructs a NON-constant LambdaScheduler(kind='incremental', max_interval=300,
  max_lambda=2) and steps it every training forward.  The loss weights live in a device tensor the loss kernels read, so a
  captured step follows a schedule that moves between replays: 20 steps of a fast ramp (interval 3), graph replay == eager bit
  for bit, and the losses are the weighted ones."""
  from mix_stage_amd.gan import IncrementalLambdaScheduler
  from mix_stage_amd.train_step import MixStageTrainStep
  import mix_stage_amd as A
  M = S = 2
  batch = [t.to(DEV) for t in O.synthetic_batch(4, M=M, S=S, seed=7)]
  audio, pose, labels, style = batch
  kinds = ['G', 'D', 'G', 'G', 'D'] * 4
  results = {}
  for use_graphs in (False, True):
    torch.manual_seed(5)
    model = _hip(M, S)
    model.lambda_scheduler = IncrementalLambdaScheduler([1.0, 1.0], max_interval=3, max_lambda=4)
    if precision == 'bf16':
      A.set_compute_dtype(model, 'bf16')
    ts = MixStageTrainStep(model, use_graphs=use_graphs)
    got, lams = [], []
    for k in kinds:
      ts.step(audio, labels, pose, style, kind=k)
      got.append([float(l) for l in ts.losses])
      lams.append((model.lambda_D, model.lambda_gan))
    results[use_graphs] = (got, lams, {n: v.clone() for n, v in model.state_dict().items()})
  eager, graph = results[False], results[True]
  assert eager[1] == graph[1]
  assert eager[1][0] == (1.0, 1.0) and eager[1][3] == (2.0, 2.0) and eager[1][-1] == (4.0, 4.0)      # the ramp, capped
  assert eager[0] == graph[0]
  for n, v in eager[2].items():
    assert torch.equal(v, graph[2][n]), n
  # the weight really multiplies the GAN term: the last G-step's generator GAN loss under lambda 4 is 4 x the unweighted one
  torch.manual_seed(5)
  plain = _hip(M, S)
  if precision == 'bf16':
    A.set_compute_dtype(plain, 'bf16')
  plain.load_state_dict({n: v for n, v in eager[2].items()})
  from test_gpu_model import _step
  plain.lambda_scheduler = IncrementalLambdaScheduler([3.0, 3.0], max_interval=10 ** 9)
  _, l3 = _step(plain, O.synthetic_batch(4, M=M, S=S, seed=7), 'G', DEV)
  plain.lambda_scheduler = IncrementalLambdaScheduler([1.0, 1.0], max_interval=10 ** 9)
  _, l1 = _step(plain, O.synthetic_batch(4, M=M, S=S, seed=7), 'G', DEV)
  assert abs(float(l3[1]) - 3.0 * float(l1[1])) <= 1e-5 * max(1.0, abs(float(l3[1]))) and abs(float(l3[0]) - float(l1[0])) <= 1e-6


def test_fp16_training_is_refused_and_bn_sync_is_per_trainer():
  """fp16 has no loss scaling here (activation gradients would underflow): the trainer refuses it instead of training
  silently wrong.  The module-level bn_sync switch follows the trainer that steps, not the one built last."""
  import mix_stage_amd as A
  from mix_stage_amd import ops
  from mix_stage_amd.train_step import MixStageTrainStep
  model = _hip(4, 4)
  A.set_compute_dtype(model, 'fp16')
  with pytest.raises(NotImplementedError):
    MixStageTrainStep(model, use_graphs=False)
  a = MixStageTrainStep(_hip(4, 4), use_graphs=False, bn_sync='global')
  b = MixStageTrainStep(_hip(4, 4), use_graphs=False, bn_sync='local')
  assert ops._bn_sync['on'] is False          # b was built last
  audio, pose, labels, style = O.synthetic_batch(4, M=4, S=4, seed=3)
  a.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind='D')
  assert ops._bn_sync['on'] is True           # ... but a's step runs with a's setting (one rank: same arithmetic as local)
  b.step(audio.to(DEV), labels.to(DEV), pose.to(DEV), style.to(DEV), kind='D')
  assert ops._bn_sync['on'] is False


def test_a_step_with_non_finite_gradients_is_refused_on_the_device():
  """A NaN anywhere in a step's gradients (here: a poisoned input; in production: the output of a launch whose in-launch meeting
  timed out) must not reach the weights, the Adam moments or the running BatchNorm statistics: ms_adam_step_segmented skips the
  update and counts it, the kernels keep the batch out of the running buffers, MixStageTrainStep reports it (per-step polling /
  check_health) -- and the next clean step trains normally."""
  import warnings
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 2
  for use_graphs in (False, True):
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=use_graphs)
    ts.on_bad_step = 'skip'
    good = [t.to(DEV) for t in O.synthetic_batch(3, M=M, S=S, seed=11)]
    audio, pose, labels, style = good
    for kind in ('G', 'D'):
      ts.step(audio, labels, pose, style, kind=kind)
    torch.cuda.synchronize()
    ts.check_health()
    assert ts.skipped_steps == 0
    snap = dict(pG=ts.optim_G.flat_p.clone(), pD=ts.optim_D.flat_p.clone(), mG=ts.optim_G.exp_avg.clone(), vG=ts.optim_G.exp_avg_sq.clone(),
                mD=ts.optim_D.exp_avg.clone(), bufs={n: b.clone() for n, b in model.named_buffers() if b.is_floating_point()})
    bad_pose = pose.clone()
    bad_pose[1, 5, 7] = float('nan')
    bad_audio = audio.clone()
    bad_audio[0, 3, 9] = float('nan')
    with warnings.catch_warnings(record=True) as caught:
      warnings.simplefilter('always')
      ts.step(bad_audio, labels, bad_pose, style, kind='G')
      ts.step(audio, labels, bad_pose, style, kind='D')
      torch.cuda.synchronize()
      ts.check_health()
    assert ts.skipped_steps == 2, ts.skipped_steps
    assert any('refused' in str(w.message) for w in caught)
    assert torch.equal(ts.optim_G.flat_p, snap['pG']) and torch.equal(ts.optim_D.flat_p, snap['pD'])
    assert torch.equal(ts.optim_G.exp_avg, snap['mG']) and torch.equal(ts.optim_G.exp_avg_sq, snap['vG']) and torch.equal(ts.optim_D.exp_avg, snap['mD'])
    for n, b in model.named_buffers():
      if b.is_floating_point():
        assert torch.isfinite(b).all(), n
        # running statistics: a poisoned batch leaves no trace.  (D's BatchNorm sees two batches per D-step, gan.py:120,126: the
        # clean fake-pose pass of the refused D-step did update D's running statistics -- finite, valid statistics.)
        if n.startswith('G.'):
          assert torch.equal(b, snap['bufs'][n]), n
    # on_bad_step = 'raise' (the default) raises instead of warning
    ts.on_bad_step = 'raise'
    ts.step(bad_audio, labels, bad_pose, style, kind='G')
    with pytest.raises(RuntimeError, match='refused'):
      ts.check_health()
    # ... and training goes on from the intact state
    ts.step(audio, labels, pose, style, kind='G')
    torch.cuda.synchronize()
    ts.check_health()
    assert not torch.equal(ts.optim_G.flat_p, snap['pG']) and torch.isfinite(ts.optim_G.flat_p).all()


def test_degrade_mode_switches_the_meetings_off_after_a_timeout_and_goes_on_training():
  """on_bad_step='degrade': a meeting that times out (here: a decoder-chain counter put out of step by hand -- what a launch that did
  not have the device to itself leaves behind) costs the steps it poisoned, then the trainer switches the in-launch meetings off,
  re-captures its steps on the per-block kernels and goes on; weights stay finite and keep moving."""
  import warnings
  from mix_stage_amd import ops, ops16
  from mix_stage_amd.train_step import MixStageTrainStep
  M = S = 4
  assert ops16.in_launch_meetings()
  try:
    model = _hip(M, S)
    ts = MixStageTrainStep(model, use_graphs=True)
    ts.on_bad_step = 'degrade'
    batch = [t.to(DEV) for t in O.synthetic_batch(4, M=M, S=S, seed=13)]
    audio, pose, labels, style = batch
    for kind in ('G', 'D', 'G'):
      ts.step(audio, labels, pose, style, kind=kind)
    ts.check_health()
    assert ts.skipped_steps == 0 and not ts.degraded
    chains = [b for k, b in ops16._bn_sync.items() if 'chain' in k]
    assert chains, 'the headline decoder runs as one chained launch'
    for b in chains:                                   # one arrival too many on the first live counter of every chain buffer
      nz = (b[ops.CHAIN_SYNC_FIRST_WORD:] != 0).nonzero()
      if len(nz):
        b[ops.CHAIN_SYNC_FIRST_WORD + int(nz[0])] += 1
    before = ts.optim_G.flat_p.clone()
    with warnings.catch_warnings(record=True) as caught:
      warnings.simplefilter('always')
      ts.step(audio, labels, pose, style, kind='G')    # a workgroup waits for an arrival that never comes: NaN, refused
      ts.check_health()
    assert ts.skipped_steps >= 1 and ts.degraded and not ops16.in_launch_meetings()
    assert any('degrade' in str(w.message) for w in caught)
    assert torch.equal(ts.optim_G.flat_p, before)      # the refused step did not touch the weights
    for kind in ('G', 'D', 'G'):
      ts.step(audio, labels, pose, style, kind=kind)
    ts.check_health()
    assert torch.isfinite(ts.optim_G.flat_p).all() and not torch.equal(ts.optim_G.flat_p, before)
    assert all(float(l) == float(l) for l in ts.losses)
  finally:
    ops16.bn_sync_clear()
    ops16.set_in_launch_meetings(True)
