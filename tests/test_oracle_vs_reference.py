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


def _n1_inputs(seed=3, B=3, T=16, P=104, M=5):
  import numpy as np
  rng = np.random.default_rng(seed)
  pose = rng.standard_normal((B, T, P)) * 30 + 100
  PK = P - 8
  centers = rng.standard_normal((M, PK + PK + PK // 2)) * 30 + 50
  mean, var = rng.standard_normal(P) * 10 + 100, rng.random(P) * 50 + 1
  var[5], var[11] = 0.0, -1e-9
  return pose, centers, mean, var


def test_prestep_oracle_vs_reference_transform():
  """N1: KMeans.get_feats / predict with the job scripts' feature list ['pose','velocity','speed'] (and the argsUtils default)
  and ZNorm.znorm, executed from the reference's own transform.py."""
  import types
  import numpy as np
  from oracle import prestep_oracle as PO
  T = refload.load_transform_and_metrics().transform
  pose, centers, mean, var = _n1_inputs()
  kept = PO.remove_joints(pose, [0, 7, 8, 9])
  for feats, width in ((['pose', 'velocity', 'speed'], 96 + 96 + 48), (['pose', 'velocity'], 192)):
    km = types.SimpleNamespace(feats=feats, centers=torch.from_numpy(centers[:, :width].copy()))
    km.get_feats = lambda x, km=km: T.KMeans.get_feats(km, x)
    f_ref = T.KMeans.get_feats(km, torch.from_numpy(kept)).numpy()
    assert np.abs(PO.kmeans_feats(kept, feats) - f_ref).max() <= 1e-12 * np.abs(f_ref).max()      # (x**0.5 vs sqrt: 1 ulp)
    l_ref = T.KMeans.predict(km, torch.from_numpy(kept)).numpy()
    assert np.array_equal(PO.kmeans_predict(kept, centers[:, :width], feats), l_ref)
  z_ref = T.ZNorm.znorm(None, torch.from_numpy(pose), [torch.from_numpy(mean), torch.from_numpy(var)]).numpy()
  assert np.abs(PO.znorm(pose, mean, var) - z_ref).max() <= 1e-12 * np.abs(z_ref).max()


def test_metrics_oracle_vs_reference_metrics():
  """N3: L1, VelL1 and PCK of the reference's own metrics.py on one batch (float64)."""
  import numpy as np
  from oracle import metrics_oracle as MO
  Mx = refload.load_transform_and_metrics().metrics
  rng = np.random.default_rng(11)
  y, gt = rng.standard_normal((4, 16, 104)), rng.standard_normal((4, 16, 104))
  mask = [0, 7, 8, 9]
  l1, vl = Mx.L1(), Mx.VelL1()
  l1(torch.from_numpy(y), torch.from_numpy(gt), mask_idx=mask); vl(torch.from_numpy(y), torch.from_numpy(gt), mask_idx=mask)
  assert abs(MO.l1(y, gt, mask) - l1.get_averages('t')['t_L1']) <= 1e-12
  assert abs(MO.vel_l1(y, gt, mask) - vl.get_averages('t')['t_VelL1']) <= 1e-12
  pck = Mx.PCK(alphas=[0.1, 0.2], num_joints=52)
  y3, g3 = y.reshape(-1, 2, 52) * 3, gt.reshape(-1, 2, 52) * 3 + 0.3 * rng.standard_normal((64, 2, 52))
  y3 = g3 + 0.4 * rng.standard_normal(g3.shape)
  pck(torch.from_numpy(y3), torch.from_numpy(g3), mask_idx=mask)
  ref = pck.get_averages('t')
  mine = MO.pck(y3, g3, mask, alphas=(0.1, 0.2))
  for a in (0.1, 0.2):
    per_joint, kept_mean = mine[a]
    for j in range(52):
      assert abs(per_joint[j] - ref['t_pck_%s_%d' % (a, j)]) <= 1e-6, (a, j)
    assert abs(kept_mean - ref['t_pck_%s' % a]) <= 1e-6


def test_fid_w1_oracle_vs_reference_metrics():
  """N3: FID and W1 of the reference's own metrics.py, fed as calculate_metrics feeds them (trainer.py:884-896) over three
  batches: running statistics, histograms and the final numbers."""
  import numpy as np
  from oracle import metrics_oracle as MO
  Mx = refload.load_transform_and_metrics().metrics
  mask = [0, 7, 8, 9]
  rng = np.random.default_rng(29)
  mean, var = rng.standard_normal(104) * 20 + 150, rng.random(104) * 400 + 25
  std = var ** 0.5
  fid, w1 = Mx.FID(), Mx.W1()
  acc = MO.EvalAccumulators(mean, var, mask)
  kept = [j for j in range(52) if j not in mask]
  for step in range(3):
    B, T = 5, 24
    gt = rng.standard_normal((B, 1, 104)) * 0.5 + np.cumsum(rng.standard_normal((B, T, 104)) * 0.08, axis=1)
    y_kept = (gt.reshape(B, T, 2, 52)[..., kept] + 0.15 * rng.standard_normal((B, T, 2, 48))).reshape(B, T, 96)
    y_kept, gt = y_kept.astype(np.float32), gt.astype(np.float32)          # the device path holds fp32 tensors
    acc.update(y_kept, gt)
    y_full = MO.reinsert_joints(y_kept.astype(np.float64), gt.astype(np.float64), mask)
    fid(torch.from_numpy(y_full), torch.from_numpy(gt.astype(np.float64)), mask_idx=mask)
    yd = torch.from_numpy((y_full * std + mean).reshape(B, T, 2, 52))
    gd = torch.from_numpy((gt.astype(np.float64) * std + mean).reshape(B, T, 2, 52))
    w1(yd, gd, mask_idx=mask)
  ref = dict(fid.get_averages('t'), **w1.get_averages('t'))
  mine = acc.averages()
  assert ref['t_FID'] != 1000 and ref['t_W1_vel'] != 1000
  assert abs(mine['FID'] - ref['t_FID']) <= 1e-9 * max(1.0, abs(ref['t_FID']))
  assert abs(mine['W1_vel'] - ref['t_W1_vel']) <= 1e-12 and abs(mine['W1_acc'] - ref['t_W1_acc']) <= 1e-12
  assert np.array_equal(acc.hist['y_vel'], w1.y_vel_meter.sum) and np.array_equal(acc.hist['gt_acc'], w1.gt_acc_meter.sum)
  assert np.abs(acc.sq['y'] - fid.y_square_meter.sum.numpy()).max() <= 1e-9 * np.abs(acc.sq['y']).max()
