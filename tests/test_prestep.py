"""N1 pre-step: oracle self-consistency on the CPU, HIP kernels vs the oracle on the GPU (labels bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import prestep_oracle as PO


def _inputs(B=6, T=64, P=104, F_=128, M=8, seed=5):
  rng = np.random.default_rng(seed)
  pose = (rng.standard_normal((B, T, P)) * 40 + 100).astype(np.float32)
  pose[:, 1:] = pose[:, :1] + np.cumsum(rng.standard_normal((B, T - 1, P)).astype(np.float32), axis=1)
  audio = rng.standard_normal((B, T, F_)).astype(np.float32) * 3 - 20
  mask = [0, 7, 8, 9]
  PK = P - 2 * len(mask)
  centers = np.concatenate([rng.standard_normal((M, PK)) * 40 + 100, rng.standard_normal((M, PK))], axis=1)
  pose_mean, pose_var = rng.standard_normal(P) * 10 + 100, rng.random(P) * 50 + 1
  pose_var[5] = 0.0                               # std == 0 -> eps (transform.py:224-225)
  pose_var[11] = -1e-9                            # negative variance is clamped (transform.py:222)
  audio_mean, audio_var = rng.standard_normal(F_), rng.random(F_) * 4 + 0.1
  return pose, audio, centers, pose_mean, pose_var, audio_mean, audio_var, mask


def test_oracle_shapes_and_edge_cases():
  pose, audio, centers, pm, pv, am, av, mask = _inputs()
  a, labels, y = PO.processed_batch(pose, audio, centers, pm, pv, am, av, mask)
  assert a.shape == audio.shape and labels.shape == pose.shape[:2] and y.shape == pose.shape[:2] + (96,)
  assert labels.dtype == np.int64 and labels.min() >= 0 and labels.max() < 8
  keep = PO.keep_columns(104, mask)
  assert len(keep) == 96 and 0 not in keep and 52 not in keep and 59 + 2 not in keep
  # velocity feature is zero at t = 0 and a tie goes to the first centre
  f = PO.kmeans_feats(PO.remove_joints(pose, mask).astype(np.float64))
  assert np.all(f[:, 0, 96:] == 0)
  twin = np.stack([centers[0], centers[0], centers[1]])
  assert set(np.unique(PO.kmeans_predict(PO.remove_joints(pose, mask), twin))) <= {0, 2}
  assert np.isfinite(y).all()


@pytest.mark.gpu
def test_hip_prestep_matches_oracle():
  from mix_stage_amd.prestep import DevicePreStep
  for seed, M in ((5, 8), (6, 25), (7, 1)):
    pose, audio, centers, pm, pv, am, av, mask = _inputs(M=M, seed=seed)
    a_ref, l_ref, y_ref = PO.processed_batch(pose, audio, centers, pm, pv, am, av, mask)
    pre = DevicePreStep(centers, pm, pv, am, av, mask=mask)
    a, labels, y = pre(torch.from_numpy(pose).cuda(), torch.from_numpy(audio).cuda())
    assert np.array_equal(labels.cpu().numpy(), l_ref)                       # index work: bit-exact
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.astype(np.float32), rtol=2e-7, atol=0)
    np.testing.assert_allclose(a.cpu().numpy(), a_ref.astype(np.float32), rtol=2e-7, atol=0)


def _fixture():
  import os
  return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'n1n3.npz'))


def test_oracle_matches_reference_vectors():
  """tests/golden/n1n3.npz holds KMeans labels (both feature lists) and ZNorm outputs computed by the reference's own
  transform.py (tests/golden/make_golden.py n1n3)."""
  z = _fixture()
  mask = [int(v) for v in z['mask']]
  kept = PO.remove_joints(z['pose'], mask)
  assert np.array_equal(PO.kmeans_predict(kept, z['centers'], ('pose', 'velocity', 'speed')), z['labels_pvs'])
  assert np.array_equal(PO.kmeans_predict(kept, z['centers'][:, :192], ('pose', 'velocity')), z['labels_pv'])
  assert np.abs(PO.znorm(z['pose'], z['mean'], z['var']) - z['znorm']).max() <= 1e-9
  assert len(np.unique(z['labels_pvs'])) > 2 and not np.array_equal(z['labels_pvs'], z['labels_pv'])


@pytest.mark.gpu
def test_hip_prestep_matches_reference_vectors():
  """The on-device pre-step with the Mix-StAGE jobs' feature list [pose, velocity, speed] (src/jobs/mix-stage.py) and with
  the argsUtils default, against labels produced by the reference: bit-exact."""
  from mix_stage_amd.prestep import DevicePreStep
  z = _fixture()
  mask = [int(v) for v in z['mask']]
  pose = torch.from_numpy(z['pose']).cuda()
  audio = torch.zeros(pose.shape[0], pose.shape[1], 4, device='cuda')
  for feats, width, key in ((('pose', 'velocity', 'speed'), 240, 'labels_pvs'), (('pose', 'velocity'), 192, 'labels_pv')):
    pre = DevicePreStep(z['centers'][:, :width], z['mean'], z['var'], np.zeros(4), np.ones(4), mask=mask, feats=feats)
    _, labels, y = pre(pose, audio)
    assert np.array_equal(labels.cpu().numpy(), z[key])
    np.testing.assert_allclose(y.cpu().numpy(), PO.remove_joints(z['znorm'], mask).astype(np.float32), rtol=3e-7, atol=0)
  with pytest.raises(ValueError):
    DevicePreStep(z['centers'], z['mean'], z['var'], np.zeros(4), np.ones(4), mask=mask)       # 240 columns need 'speed'
