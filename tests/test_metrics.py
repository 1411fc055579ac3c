"""N3 step metrics: oracle sanity on the CPU, HIP kernel vs the oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO


def _inputs(B=5, T=64, P=104, seed=3):
  rng = np.random.default_rng(seed)
  mask = [0, 7, 8, 9]
  gt = rng.standard_normal((B, T, P)).astype(np.float32)
  ycap = (gt.reshape(B, T, 2, 52)[..., [j for j in range(52) if j not in mask]].reshape(B, T, 96)
          + 0.15 * rng.standard_normal((B, T, 96))).astype(np.float32)
  mean, var = rng.standard_normal(P) * 20 + 100, rng.random(P) * 400 + 50
  return ycap, gt, mean, var, mask


def test_oracle_identities():
  ycap, gt, mean, var, mask = _inputs()
  perfect = gt.reshape(5, 64, 2, 52)[..., [j for j in range(52) if j not in mask]].reshape(5, 64, 96)
  r = MO.step_metrics(perfect, gt, mean, var, mask)
  assert r['L1'] == 0 and r['VelL1'] == 0 and r['pck'][0.1][1] == 1.0
  r = MO.step_metrics(ycap, gt, mean, var, mask)
  assert 0.05 < r['L1'] < 0.2 and r['VelL1'] > r['L1']
  assert r['pck'][0.1][1] <= r['pck'][0.2][1] <= 1.0
  assert np.all(r['pck'][0.1][0][mask] == 1.0)        # re-inserted joints coincide with the ground truth


@pytest.mark.gpu
def test_hip_metrics_match_oracle():
  from mix_stage_amd.metrics import DeviceStepMetrics
  ycap, gt, mean, var, mask = _inputs()
  ref = MO.step_metrics(ycap, gt, mean, var, mask)
  m = DeviceStepMetrics(mean, var, mask=mask)
  got = m.update(torch.from_numpy(ycap).cuda(), torch.from_numpy(gt).cuda())
  assert abs(got['L1'] - ref['L1']) < 1e-9 and abs(got['VelL1'] - ref['VelL1']) < 1e-9
  for a in (0.1, 0.2):
    np.testing.assert_allclose(got['pck'][a][0].numpy(), ref['pck'][a][0], atol=1e-12)     # hit counts: exact
    assert abs(got['pck'][a][1] - ref['pck'][a][1]) < 1e-12
  avg = m.averages('train')
  assert abs(avg['train_L1'] - ref['L1']) < 1e-9 and abs(avg['train_pck_0.1'] - ref['pck'][0.1][1]) < 1e-12


def test_oracle_matches_reference_vectors():
  """L1, VelL1 and PCK of tests/golden/n1n3.npz were computed by the reference's own metrics.py (make_golden.py n1n3)."""
  import os
  from oracle import metrics_oracle as MO
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'n1n3.npz'))
  mask = [int(v) for v in z['mask']]
  assert abs(MO.l1(z['m_y'], z['m_gt'], mask) - float(z['L1'])) <= 1e-12
  assert abs(MO.vel_l1(z['m_y'], z['m_gt'], mask) - float(z['VelL1'])) <= 1e-12
  res = MO.pck(z['m_y'].reshape(-1, 2, 52), z['m_gt'].reshape(-1, 2, 52), mask, alphas=(0.1, 0.2))
  for i, a in enumerate((0.1, 0.2)):
    assert np.abs(res[a][0] - z['pck'][i]).max() <= 1e-6 and abs(res[a][1] - z['pck_mean'][i]) <= 1e-6


def _evalacc_golden(golden_dir):
  import os
  return np.load(os.path.join(golden_dir, 'evalacc.npz'))


def test_fid_w1_oracle_matches_reference_fixture(golden_dir):
  """FID / W1: the oracle against numbers, histograms and Gram matrices the reference's own metrics.py produced over three
  batches (tests/golden/make_golden.py evalacc)."""
  z = _evalacc_golden(golden_dir)
  acc = MO.EvalAccumulators(z['mean'], z['var'], list(z['mask']))
  for y, gt in zip(z['y'], z['gt']):
    acc.update(y, gt)
  got = acc.averages()
  assert abs(got['FID'] - float(z['FID'])) <= 1e-9 * max(1.0, abs(float(z['FID'])))
  assert abs(got['W1_vel'] - float(z['W1_vel'])) <= 1e-12 and abs(got['W1_acc'] - float(z['W1_acc'])) <= 1e-12
  hist = np.stack([acc.hist['y_vel'], acc.hist['y_acc'], acc.hist['gt_vel'], acc.hist['gt_acc']])
  assert np.array_equal(hist, z['hist'])
  np.testing.assert_allclose(acc.sq['gt'], z['gt_square'], rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_hip_fid_w1_accumulators_match_reference_fixture(golden_dir):
  """The device accumulators over the same three batches: histograms bit-exact, sums / Gram matrices to fp64 rounding, final
  FID / W1 equal to the reference's numbers."""
  from mix_stage_amd.metrics import DeviceEvalAccumulators
  z = _evalacc_golden(golden_dir)
  dev = DeviceEvalAccumulators(z['mean'], z['var'], mask=tuple(int(v) for v in z['mask']))
  for y, gt in zip(z['y'], z['gt']):
    dev.update(torch.from_numpy(y).cuda(), torch.from_numpy(gt).cuda())
  h = dev.w1_hist.cpu().numpy()                     # [prediction | gt][speed | acceleration]
  assert np.array_equal(np.stack([h[0, 0], h[0, 1], h[1, 0], h[1, 1]]), z['hist'])
  np.testing.assert_allclose(dev.fid_sums.cpu().numpy()[0], z['y_sum'][0], rtol=1e-12, atol=1e-12)
  np.testing.assert_allclose(dev.fid_gram.cpu().numpy()[0], z['y_square'], rtol=1e-12, atol=1e-11)
  np.testing.assert_allclose(dev.fid_gram.cpu().numpy()[1], z['gt_square'], rtol=1e-12, atol=1e-11)
  got = dev.averages('t')
  assert abs(got['t_FID'] - float(z['FID'])) <= 1e-8 * max(1.0, abs(float(z['FID'])))
  assert abs(got['t_W1_vel'] - float(z['W1_vel'])) <= 1e-12 and abs(got['t_W1_acc'] - float(z['W1_acc'])) <= 1e-12
  dev.reset()
  assert int(dev.w1_hist.sum()) == 0 and dev.rows == 0
