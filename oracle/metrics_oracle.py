"""CPU ORACLE (test infrastructure, not product code) for the per-step quality metrics that the reference computes on the
CPU after every step ("next" row N3 of SURVEY.md section 8f), restated from /root/reference/src:

  evaluation/metrics.py:94-109    L1      mean |y - gt| over the kept joints of the (B,T,2,J) view
  evaluation/metrics.py:111-131   VelL1   the same on first differences along time
  evaluation/metrics.py:247-303   PCK     per alpha: dist = ||y - gt||_2 per joint < alpha * max(h, w) of the gt pose
  model/trainer.py:865-915        calculate_metrics glue: re-insert the removed joints, L1 / VelL1 on the normalised
                                  poses, undo the normalisation, put the root joint at (0,0), PCK on (B*T, 2, J)
  data/transform.py:228-229       ZNorm.inv_znorm  x * var**0.5 + mean

  evaluation/metrics.py:374-473   FID     running sums / Gram matrices of the kept columns, Frechet distance of the Gaussians
  evaluation/metrics.py:476-532   W1      histograms of the per-frame mean joint speed / acceleration, Wasserstein-1

PINNED for L1, VelL1, PCK, FID and W1: tests/test_oracle_vs_reference.py runs the reference's own metrics.py (behind the stubs of
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


class EvalAccumulators:
  """FID (metrics.py:374-473) and W1 (:476-532) restated: the same running quantities the reference keeps in its AverageMeter
  objects (sum += val * n), the same final formulas.  update() takes what calculate_metrics (trainer.py:865-896) hands them:
  the normalised prediction with its removed joints re-inserted (FID) and the de-normalised poses (W1)."""

  def __init__(self, mean, var, mask=(0, 7, 8, 9), bin_width=0.1, max_value=300.0):
    self.mean, self.std = np.asarray(mean, np.float64), np.asarray(var, np.float64) ** 0.5
    self.mask = list(mask)
    self.ranges = np.arange(0, max_value, bin_width)
    self.n = 0
    self.sum = {k: 0 for k in ('y', 'gt')}
    self.sq = {k: 0 for k in ('y', 'gt')}
    self.hist = {k: 0 for k in ('y_vel', 'y_acc', 'gt_vel', 'gt_acc')}

  @staticmethod
  def _vel_acc(poses):
    """(B,T,2,J) -> per frame transition the joint speeds' mean, flattened; the same for second differences (W1.get_vel_acc)."""
    def mean_joint_norm(d):                                   # d: (B, T', 2, J) -> (B*T',)
      return np.sqrt(np.square(d).sum(axis=2)).mean(axis=-1).ravel()
    first = np.diff(poses, n=1, axis=1)
    second = np.diff(first, n=1, axis=1)
    return mean_joint_norm(first), mean_joint_norm(second)

  def update(self, y_cap_kept, gt_full_norm):
    y_cap_kept, gt = y_cap_kept.astype(np.float64), gt_full_norm.astype(np.float64)
    y_full = reinsert_joints(y_cap_kept, gt, self.mask)
    B, T, P = gt.shape
    J = P // 2
    kept = sorted(set(range(J)) - set(self.mask))
    rows = {'y': y_full.reshape(B, T, 2, J)[..., kept].reshape(B * T, -1), 'gt': gt.reshape(B, T, 2, J)[..., kept].reshape(B * T, -1)}
    n = B * T
    self.n += n
    for k, v in rows.items():                                 # FID.__call__
      self.sum[k] = self.sum[k] + v.mean(0, keepdims=True) * n
      self.sq[k] = self.sq[k] + (v.T @ v / n) * n
    for k, v in (('y', y_full), ('gt', gt)):                  # W1.__call__ on the de-normalised poses
      d = (v * self.std + self.mean).reshape(B, T, 2, J)[..., kept]
      vel, acc = self._vel_acc(d)
      self.hist[k + '_vel'] = self.hist[k + '_vel'] + np.histogram(vel, bins=self.ranges)[0]
      self.hist[k + '_acc'] = self.hist[k + '_acc'] + np.histogram(acc, bins=self.ranges)[0]

  def statistics(self):
    out = {}
    for k in ('y', 'gt'):
      s, N = self.sum[k], self.n
      out[k] = ((s / N).squeeze(), (self.sq[k] - s.T @ s / N) / (N - 1))
    return out

  def averages(self):
    import scipy.stats
    from scipy import linalg
    st = self.statistics()
    (mu1, s1), (mu2, s2) = st['gt'], st['y']                  # calculate_frechet_distance(gt_mu, gt_sigma, y_mu, y_sigma)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(s1.dot(s2), disp=False)
    if not np.isfinite(covmean).all():
      off = np.eye(s1.shape[0]) * 1e-6
      covmean = linalg.sqrtm((s1 + off).dot(s2 + off))
    if np.iscomplexobj(covmean):
      covmean = covmean.real
    fid = float(diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean))
    N = self.ranges[:-1]
    return dict(FID=fid, W1_vel=scipy.stats.wasserstein_distance(N, N, self.hist['y_vel'], self.hist['gt_vel']),
                W1_acc=scipy.stats.wasserstein_distance(N, N, self.hist['y_acc'], self.hist['gt_acc']))
