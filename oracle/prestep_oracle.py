"""CPU ORACLE (test infrastructure, not product code) for the per-step pre-processing in front of the GAN path
("next" row N1 of SURVEY.md section 8f), restated from /root/reference/src:

  data/transform.py:352-376  KMeans.get_feats   ('pose' and 'velocity' features)
  data/transform.py:395-410  KMeans.predict     (fp64 squared distance to the centres, first-minimum argmin)
  data/transform.py:221-226  ZNorm.znorm        ((x - mean) / std, std = sqrt(max(var,0)), std == 0 -> eps)
  data/transform.py:481-507  RemoveJoints       (drop the masked joints from the (.., 2, J) view)
  model/trainer.py:1290-1308 get_processed_batch (labels from the RAW pose, then znorm, then RemoveJoints)

PINNED for KMeans.get_feats / predict (all of 'pose', 'velocity', 'speed') and ZNorm.znorm: tests/test_oracle_vs_reference.py
runs the reference's own transform.py (oracle/refload.load_transform_and_metrics stubs the dataset-stack modules it imports)
and tests/golden/n1n3.npz holds vectors it produced.  PARITY UNPINNED for RemoveJoints only: it delegates to
pycasper.torchUtils.remove_slices, which is not in the reference tree; "remove the listed indices along the last axis"
is inferred from the call site and the shapes (104 -> 96 features for mask [0,7,8,9], trainer.py:1353).
"""
import numpy as np


def keep_columns(num_feats, mask):
  """Columns of the flat (x-block | y-block) pose vector that survive RemoveJoints(mask)."""
  J = num_feats // 2
  kept = [j for j in range(J) if j not in set(mask)]
  return np.array([xy * J + j for xy in range(2) for j in kept], dtype=np.int64)


def remove_joints(pose, mask):
  return pose[..., keep_columns(pose.shape[-1], mask)]


def kmeans_feats(x, feats=('pose', 'velocity')):
  """transform.py:352-378: 'pose' | 'velocity' | 'speed' blocks in the order of `feats`."""
  v = np.zeros_like(x)
  v[:, 1:, :] = x[:, 1:] - x[:, :-1]
  out = []
  for f in feats:
    if f == 'pose':
      out.append(x)
    elif f == 'velocity':
      out.append(v)
    elif f == 'speed':
      vv = v.reshape(v.shape[0], v.shape[1], 2, -1)
      out.append(np.sqrt((vv ** 2).sum(axis=-2)))
    else:
      raise ValueError(f)
  return np.concatenate(out, axis=-1)


def kmeans_predict(pose_removed, centers, feats=('pose', 'velocity')):
  f = kmeans_feats(pose_removed.astype(np.float64), feats)
  B, T, D = f.shape
  mse = ((centers[None, :, :].astype(np.float64) - f.reshape(-1, 1, D)) ** 2).sum(-1)
  return mse.argmin(-1).reshape(B, T).astype(np.int64)          # first minimum, like torch.min(dim)[1]


def znorm(x, mean, var, eps=1e-8):
  var = np.asarray(var, dtype=np.float64)
  std = np.sqrt(var * (var >= 0))
  std = np.where(std == 0, eps, std)
  return (x.astype(np.float64) - mean) / std


def processed_batch(pose_raw, audio_raw, centers, pose_mean, pose_var, audio_mean, audio_var, mask, feats=('pose', 'velocity')):
  """-> (audio_norm, labels, y) as TrainerLateClusterGAN.get_processed_batch hands them to the model."""
  labels = kmeans_predict(remove_joints(pose_raw, mask), centers, feats)
  y = remove_joints(znorm(pose_raw, pose_mean, pose_var), mask)
  return znorm(audio_raw, audio_mean, audio_var), labels, y
