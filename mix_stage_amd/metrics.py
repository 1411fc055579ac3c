"""Per-step quality metrics on the MI355X ("next" row N3 of SURVEY.md section 8f): L1, VelL1 and PCK.

Reference: TrainerBase.calculate_metrics (src/model/trainer.py:865-915) copies y_cap to the CPU after EVERY step and runs
evaluation.metrics.{L1, VelL1, PCK, ...} there (metrics.py:94-131,247-303).  Here one kernel produces the numerators per
clip from the device-resident prediction; the running averages are kept on the host exactly like the reference's
AverageMeter objects (weights n = B, resp. n = B*T*len(kept) for PCK).  FID / W1 / diversity / F1 stay out of scope.
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
