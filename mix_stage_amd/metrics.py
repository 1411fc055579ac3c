"""Per-step quality metrics on the MI355X ("next" row N3 of SURVEY.md section 8f): L1, VelL1 and PCK.

Reference: TrainerBase.calculate_metrics (src/model/trainer.py:865-915) copies y_cap to the CPU after EVERY step and runs
evaluation.metrics.{L1, VelL1, PCK, ...} there (metrics.py:94-131,247-303).  Here one kernel produces the numerators per
clip from the device-resident prediction; the running averages are kept on the host exactly like the reference's
AverageMeter objects (weights n = B, resp. n = B*T*len(kept) for PCK).  DeviceEvalAccumulators keeps the FID sufficient
statistics and the W1 speed / acceleration histograms (metrics.py:374-532) in device buffers across steps; only the final
reduction (matrix square root, Wasserstein distance: scipy, as in the reference) runs on the host.  Diversity / expressiveness
/ cluster F1 stay out of scope.
"""
import ctypes

import torch

from ._lib import check, lib

_vp = ctypes.c_void_p


def _p(t):
  return _vp(t.data_ptr())


class DeviceStepMetrics:
  def __init__(self, pose_mean, pose_var, mask=(0, 7, 8, 9), num_feats=104, alphas=(0.1, 0.2), device='cuda:0'):
    dev = torch.device(device)
    J = num_feats // 2
    kept = [j for j in range(J) if j not in set(mask)]
    keep = [xy * J + j for xy in range(2) for j in kept]
    slot = [-1] * num_feats
    for i, c in enumerate(keep):
      slot[c] = i
    self.P, self.PK, self.J, self.kept = num_feats, len(keep), J, kept
    self.keep = torch.tensor(keep, dtype=torch.int32, device=dev)
    self.slot_of = torch.tensor(slot, dtype=torch.int32, device=dev)
    self.mean = torch.as_tensor(pose_mean, dtype=torch.float64).reshape(-1).to(dev)
    self.std = (torch.as_tensor(pose_var, dtype=torch.float64).reshape(-1) ** 0.5).to(dev)     # transform.py:228-229
    self.alphas = tuple(alphas)
    self.alphas_dev = torch.tensor(alphas, dtype=torch.float32, device=dev)
    self.dev = dev
    self.reset()

  def reset(self):
    self.count = 0
    self.l1_sum = self.vel_sum = 0.0
    self.pck_joint = {a: torch.zeros(self.J, dtype=torch.float64) for a in self.alphas}
    self.pck_mask = {a: [0.0, 0] for a in self.alphas}

  def batch_numerators(self, y_cap, gt_full_norm):
    """y_cap (B,T,PK) fp32 prediction, gt_full_norm (B,T,P) fp32 normalised ground truth, both on the device.
    Returns the (B, 2 + n_alpha*J) fp64 device tensor (no host sync)."""
    if not (y_cap.is_cuda and gt_full_norm.is_cuda):
      raise TypeError('DeviceStepMetrics runs on the MI355X')
    y_cap, gt = y_cap.detach().contiguous(), gt_full_norm.contiguous()
    B, T, PK = y_cap.shape
    assert PK == self.PK and gt.shape == (B, T, self.P)
    out = torch.empty((B, 2 + len(self.alphas) * self.J), dtype=torch.float64, device=self.dev)
    check(lib().ms_step_metrics(_p(y_cap), _p(gt), _p(self.keep), _p(self.slot_of), _p(self.mean), _p(self.std),
                                _p(self.alphas_dev), len(self.alphas), _p(out), B, T, self.P, PK,
                                _vp(torch.cuda.current_stream().cuda_stream)), 'ms_step_metrics')
    return out

  def update(self, y_cap, gt_full_norm):
    """Accumulate one batch into the running averages (one small D2H copy of the numerators)."""
    B, T = y_cap.shape[:2]
    num = self.batch_numerators(y_cap, gt_full_norm).sum(0).cpu()
    batch = dict(L1=float(num[0]) / (B * T * self.PK), VelL1=float(num[1]) / (B * (T - 1) * self.PK), pck={})
    self.l1_sum += batch['L1'] * B
    self.vel_sum += batch['VelL1'] * B
    self.count += B
    for i, a in enumerate(self.alphas):
      hits = num[2 + i * self.J:2 + (i + 1) * self.J] / (B * T)
      self.pck_joint[a] += hits * (B * T)
      m = float(hits[self.kept].mean())
      self.pck_mask[a][0] += m * (B * T * len(self.kept))
      self.pck_mask[a][1] += B * T * len(self.kept)
      batch['pck'][a] = (hits, m)
    return batch

  def averages(self, desc='train'):
    out = {'%s_L1' % desc: self.l1_sum / max(1, self.count), '%s_VelL1' % desc: self.vel_sum / max(1, self.count)}
    for a in self.alphas:
      out['%s_pck_%s' % (desc, a)] = self.pck_mask[a][0] / max(1, self.pck_mask[a][1])
    return out


class DeviceEvalAccumulators:
  """FID and W1 of evaluation/metrics.py (classes FID :374-473, W1 :476-532) with the per-step accumulation on the device:
  `update` adds one batch to running fp64 sums / Gram matrices / integer histograms without any host synchronisation;
  `averages` copies them once and finishes on the host exactly like FID.get_averages / W1.get_averages."""

  def __init__(self, pose_mean, pose_var, mask=(0, 7, 8, 9), num_feats=104, bin_width=0.1, max_value=300.0, device='cuda:0'):
    dev = torch.device(device)
    J = num_feats // 2
    kept = [j for j in range(J) if j not in set(mask)]
    keep = [xy * J + j for xy in range(2) for j in kept]
    self.P, self.PK = num_feats, len(keep)
    self.keep = torch.tensor(keep, dtype=torch.int32, device=dev)
    self.mean = torch.as_tensor(pose_mean, dtype=torch.float64).reshape(-1).to(dev)
    self.std = (torch.as_tensor(pose_var, dtype=torch.float64).reshape(-1) ** 0.5).to(dev)
    import numpy as np
    self.edges = np.arange(0, max_value, bin_width)            # metrics.py:482
    self.bin_width, self.nbins = float(bin_width), len(self.edges) - 1
    self.dev = dev
    self.reset()

  def reset(self):
    self.rows = 0
    self.fid_sums = torch.zeros((2, self.PK), dtype=torch.float64, device=self.dev)
    self.fid_gram = torch.zeros((2, self.PK, self.PK), dtype=torch.float64, device=self.dev)
    self.w1_hist = torch.zeros((2, 2, self.nbins), dtype=torch.int64, device=self.dev)

  def update(self, y_cap, gt_full_norm):
    """y_cap (B,T,PK) fp32 normalised prediction (kept joints), gt_full_norm (B,T,P) fp32 normalised ground truth."""
    if not (y_cap.is_cuda and gt_full_norm.is_cuda):
      raise TypeError('DeviceEvalAccumulators runs on the MI355X')
    y_cap, gt = y_cap.detach().contiguous(), gt_full_norm.contiguous()
    B, T, PK = y_cap.shape
    assert PK == self.PK and gt.shape == (B, T, self.P) and y_cap.dtype == torch.float32 and gt.dtype == torch.float32
    check(lib().ms_eval_accumulate(_p(y_cap), _p(gt), _p(self.keep), _p(self.mean), _p(self.std), _p(self.fid_sums),
                                   _p(self.fid_gram), _p(self.w1_hist), B, T, self.P, PK, self.bin_width, self.nbins,
                                   _vp(torch.cuda.current_stream().cuda_stream)), 'ms_eval_accumulate')
    self.rows += B * T

  def statistics(self):
    """(mu, sigma) of prediction and ground truth as FID.get_averages forms them (metrics.py:448-466), and the histograms."""
    N = self.rows
    sums, gram = self.fid_sums.cpu().numpy(), self.fid_gram.cpu().numpy()
    out = {}
    for i, name in enumerate(('y', 'gt')):
      mu = sums[i] / N
      sigma = (gram[i] - sums[i][:, None] * sums[i][None, :] / N) / (N - 1)
      out[name] = (mu, sigma)
    out['hist'] = self.w1_hist.cpu().numpy()
    return out

  def averages(self, desc='train'):
    import numpy as np
    import scipy.stats
    from scipy import linalg
    st = self.statistics()
    (mu_y, sig_y), (mu_g, sig_g) = st['y'], st['gt']
    try:                                                    # calculate_frechet_distance(gt, y), metrics.py:396-446
      diff = mu_g - mu_y
      covmean, _ = linalg.sqrtm(sig_g.dot(sig_y), disp=False)
      if not np.isfinite(covmean).all():
        off = np.eye(sig_g.shape[0]) * 1e-6
        covmean = linalg.sqrtm((sig_g + off).dot(sig_y + off))
      if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
          raise ValueError('imaginary component')
        covmean = covmean.real
      fid = float(diff.dot(diff) + np.trace(sig_g) + np.trace(sig_y) - 2 * np.trace(covmean))
    except Exception:                                       # the reference reports 1000 on any failure (metrics.py:467-468)
      fid = 1000
    centers = self.edges[:-1]
    h = st['hist']
    try:
      w1_vel = scipy.stats.wasserstein_distance(centers, centers, h[0, 0], h[1, 0])
      w1_acc = scipy.stats.wasserstein_distance(centers, centers, h[0, 1], h[1, 1])
    except Exception:
      w1_vel = w1_acc = 1000
    return {'%s_FID' % desc: fid, '%s_W1_vel' % desc: w1_vel, '%s_W1_acc' % desc: w1_acc}
