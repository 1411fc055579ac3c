"""Container-only: the oracle restatement against the reference's own model files, imported
read-only from /root/reference (skipped where that tree does not exist, e.g. the GPU box)."""
import pytest
import torch

from oracle import mixstage_oracle as O
from oracle import refload

pytestmark = pytest.mark.skipif(not refload.available(), reason='reference tree not present')


@pytest.mark.parametrize('M,S,dtype,tol', [(2, 2, torch.float64, 1e-12), (3, 5, torch.float32, 1e-6)])
def test_outputs_losses_and_all_gradients(M, S, dtype, tol):
  om = O.build_gan(M=M, S=S, dtype=dtype)
  rm = refload.build_ref_gan(M=M, S=S, dtype=dtype, state=om.state_dict())
  assert list(om.state_dict().keys()) == list(rm.state_dict().keys())
  audio, pose, labels, style = O.synthetic_batch(4, M=M, S=S, dtype=dtype)
  for kind in ('G', 'D'):
    outs = []
    for m in (om, rm):
      m.train(); m.zero_grad()
      m.D_prob = 1.1 if kind == 'D' else -1.0
      torch.manual_seed(3)
      fake, losses, _ = m([audio, labels], pose, **O.model_kwargs(style))
      sum(losses).backward()
      outs.append((fake, losses))
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= tol
    for a, b in zip(outs[0][1], outs[1][1]):
      assert abs(float(a) - float(b)) <= tol
    for (n, p), (n2, q) in zip(om.named_parameters(), rm.named_parameters()):
      assert n == n2 and (p.grad is None) == (q.grad is None), n
      if p.grad is not None:
        assert (p.grad - q.grad).abs().max().item() <= tol * max(1.0, q.grad.abs().max().item()), n
    for (k, a), (_, b) in zip(om.state_dict().items(), rm.state_dict().items()):
      assert (a.double() - b.double()).abs().max().item() <= tol, k


def test_rng_consumption_matches():
  """Two host draws per GAN.forward in train mode (gan.py:105, joint_late...:127)."""
  om = O.build_gan(M=2, S=2)
  rm = refload.build_ref_gan(M=2, S=2, state=om.state_dict())
  audio, pose, labels, style = O.synthetic_batch(2, M=2, S=2)
  after = []
  for m in (om, rm):
    m.train()
    m.D_prob = 0.5
    torch.manual_seed(11)
    m([audio, labels], pose, **O.model_kwargs(style))
    after.append((torch.rand(1).item(), m.G_flag))
  assert after[0] == after[1]


def test_oracle_pose_branch_vs_reference():
  """Row a11: a fresh curriculum (thresh = 0) sends the first training calls through PoseEncoder(y) instead of the audio
  encoder (joint_late_cluster_soft_style.py:127-129, layers.py:677-696).  refload.build_ref_gan pins the audio branch, so
  this is the only place the oracle's pose branch meets the reference's."""
  dtype = torch.float64
  om = O.build_gan(M=2, S=3, dtype=dtype)
  rm = refload.build_ref_gan(M=2, S=3, dtype=dtype, state=om.state_dict())
  audio, pose, labels, style = O.synthetic_batch(3, M=2, S=3, dtype=dtype)
  for step in range(2):
    outs = []
    for m in (om, rm):
      if step == 0:
        m.G.thresh.value, m.G.thresh.iters = 0.0, 0
      m.train(); m.zero_grad()
      m.D_prob = -1.0
      torch.manual_seed(21 + step)
      fake, losses, _ = m([audio, labels], pose, **O.model_kwargs(style))
      sum(losses).backward()
      outs.append((fake, losses))
    assert om.G.thresh.value == rm.G.thresh.value == (step + 1) / 1000
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-12
    for a, b in zip(outs[0][1], outs[1][1]):
      assert abs(float(a) - float(b)) <= 1e-12
    for (n, p), (_, q) in zip(om.named_parameters(), rm.named_parameters()):
      assert (p.grad is None) == (q.grad is None), n
      if p.grad is not None:
        assert (p.grad - q.grad).abs().max().item() <= 1e-12 * max(1.0, q.grad.abs().max().item()), n
    # the pose branch trains pose_encoder and leaves audio_encoder untouched
    assert rm.G.pose_encoder.conv[0].conv.weight.grad is not None
    assert rm.G.audio_encoder.conv[0].conv.weight.grad is None
