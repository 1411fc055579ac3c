"""CPU ORACLE (test infrastructure, not product code) for the per-step quality metrics that the reference computes on the
CPU after every step ("next" row N3 of SURVEY.md section 8f), restated from /root/reference/src:

  evaluation/metrics.py:94-109    L1      mean |y - gt| over the kept joints of the (B,T,2,J) view
  evaluation/metrics.py:111-131   VelL1   the same on first differences along time
  evaluation/metrics.py:247-303   PCK     per alpha: dist = ||y - gt||_2 per joint < alpha * max(h, w) of the gt pose
  model/trainer.py:865-915        calculate_metrics glue: re-insert the removed joints, L1 / VelL1 on the normalised
                                  poses, undo the normalisation, put the root joint at (0,0), PCK on (B*T, 2, J)
  data/transform.py:228-229       ZNorm.inv_znorm  x * var**0.5 + mean

PINNED for L1, VelL1 and PCK: tests/test_oracle_vs_reference.py runs the reference's own metrics.py (behind the stubs of
oracle/refload.load_transform_and_metrics) and tests/golden/n1n3.npz holds values it produced.  PARITY UNPINNED for the
glue around them (reinsert_joints / step_metrics): RemoveJoints(inv=True) delegates to pycasper.torchUtils.add_slices;
"masked joints take the ground truth's values" (parents=None) is inferred from transform.py:483-497.
"""
import numpy as np


def reinsert_joints(y_cap_kept, gt_full, mask):
  """(B,T,2*(J-len(mask))) prediction -> (B,T,2*J): the removed joints are filled from the ground truth."""
  J = gt_full.shape[-1] // 2
  kept = [j for j in range(J) if j not in set(mask)]
  out = gt_full.reshape(gt_full.shape[0], gt_full.shape[1], 2, J).copy()
  out[..., kept] = y_cap_kept.reshape(y_cap_kept.shape[0], y_cap_kept.shape[1], 2, len(kept))
  return out.reshape(gt_full.shape)


def l1(y, gt, mask):
  J = y.shape[-1] // 2
  kept = sorted(set(range(J)) - set(mask))
  y4, g4 = y.reshape(y.shape[0], y.shape[1], 2, J), gt.reshape(gt.shape[0], gt.shape[1], 2, J)
  return np.abs(y4[..., kept] - g4[..., kept]).mean()


def vel_l1(y, gt, mask):
  J = y.shape[-1] // 2
  kept = sorted(set(range(J)) - set(mask))
  y4, g4 = y.reshape(y.shape[0], y.shape[1], 2, J), gt.reshape(gt.shape[0], gt.shape[1], 2, J)
  return np.abs((y4[:, 1:] - y4[:, :-1])[..., kept] - (g4[:, 1:] - g4[:, :-1])[..., kept]).mean()


def pck(y, gt, mask, alphas=(0.1, 0.2)):
  """y, gt: (N, 2, J).  Returns {alpha: (per-joint mean (J,), mean over the kept joints)}."""
  J = y.shape[-1]
  kept = sorted(set(range(J)) - set(mask))
  dist = np.sqrt(((y - gt) ** 2).sum(axis=1))                     # (N, J)
  out = {}
  for a in alphas:
    h = gt[:, 0, :].max(-1) - gt[:, 0, :].min(-1)
    w = gt[:, 1, :].max(-1) - gt[:, 1, :].min(-1)
    thresh = a * np.maximum(h, w)[:, None]
    hit = (dist < thresh).astype(np.float64)
    out[a] = (hit.mean(0), hit[:, kept].mean())
  return out


def step_metrics(y_cap_kept, gt_full_norm, mean, var, mask, alphas=(0.1, 0.2)):
  """The L1 / VelL1 / PCK part of TrainerBase.calculate_metrics for one batch (float64)."""
  y_cap_kept, gt = y_cap_kept.astype(np.float64), gt_full_norm.astype(np.float64)
  y_full = reinsert_joints(y_cap_kept, gt, mask)
  res = dict(L1=l1(y_full, gt, mask), VelL1=vel_l1(y_full, gt, mask))
  std = np.asarray(var, dtype=np.float64) ** 0.5
  J = gt.shape[-1] // 2
  yd = (y_full * std + mean).reshape(-1, 2, J).copy()
  gd = (gt * std + mean).reshape(-1, 2, J).copy()
  yd[..., 0] = 0
  gd[..., 0] = 0
  res['pck'] = pck(yd, gd, mask, alphas)
  return res
