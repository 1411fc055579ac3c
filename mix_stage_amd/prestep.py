"""Per-step pre-processing in front of the GAN path, on the MI355X ("next" row N1 of SURVEY.md section 8f).

Reference: TrainerLateClusterGAN.get_processed_batch (src/model/trainer.py:1290-1308) runs, on the CPU in float64 every
step: KMeans.predict on RemoveJoints(pose) -> cluster labels (src/data/transform.py:352-410), ZNorm of audio and pose
(transform.py:221-226), RemoveJoints of the normalised pose (transform.py:481-507), then copies everything to the
device.  Here the raw batch is copied once and the three transforms are two HIP kernels (fp64 arithmetic inside, like
the reference; fp32 tensors out for the fp32 model).  Loading the k-means centres / mean-variance files (HDF5) stays with
the caller; RemoveJoints follows the inferred `remove_slices` semantics (pycasper is not in the reference tree).
"""
import ctypes

import torch

from ._lib import check, lib

_vp = ctypes.c_void_p


def _p(t):
  return None if t is None else _vp(t.data_ptr())


class DevicePreStep:
  FEAT_BITS = {'pose': 1, 'velocity': 2, 'speed': 4}

  def __init__(self, centers, pose_mean, pose_var, audio_mean, audio_var, mask=(0, 7, 8, 9), num_feats=104, eps=1e-8,
               device='cuda:0', feats=('pose', 'velocity')):
    """feats: the k-means feature blocks (KMeans.get_feats, transform.py:352-378): argsUtils' default is
    ['pose', 'velocity']; the Mix-StAGE job scripts (src/jobs/mix-stage.py) train with ['pose', 'velocity', 'speed'].
    'acceleration' and 'spatial' are not offered."""
    dev = torch.device(device)
    J = num_feats // 2
    kept = [j for j in range(J) if j not in set(mask)]
    keep = [xy * J + j for xy in range(2) for j in kept]
    self.P, self.PK = num_feats, len(keep)
    self.keep = torch.tensor(keep, dtype=torch.int32, device=dev)
    feats = tuple(feats)
    unknown = [f for f in feats if f not in self.FEAT_BITS]
    if unknown or list(feats) != [f for f in ('pose', 'velocity', 'speed') if f in feats]:
      raise ValueError('feats must be a sub-list of [pose, velocity, speed] in that order, got %s' % (feats,))
    self.feats = sum(self.FEAT_BITS[f] for f in feats)
    width = sum({'pose': self.PK, 'velocity': self.PK, 'speed': self.PK // 2}[f] for f in feats)
    centers = torch.as_tensor(centers, dtype=torch.float64)
    if centers.shape[1] != width:
      raise ValueError('centres must have %d columns (%s of the kept joints), got %d' % (width, ' | '.join(feats), centers.shape[1]))
    self.M = centers.shape[0]
    self.centers = centers.contiguous().to(dev)

    def prep(mean, var):
      mean = torch.as_tensor(mean, dtype=torch.float64).reshape(-1)
      var = torch.as_tensor(var, dtype=torch.float64).reshape(-1)
      std = (var * (var >= 0)).sqrt()
      std = torch.where(std == 0, torch.full_like(std, eps), std)      # transform.py:222-225
      return mean.to(dev), (1.0 / std).to(dev)
    self.pose_mean, self.pose_inv = prep(pose_mean, pose_var)
    self.audio_mean, self.audio_inv = prep(audio_mean, audio_var)
    self.dev = dev

  def __call__(self, pose_raw, audio_raw):
    """pose_raw (B,T,P) and audio_raw (B,T,F) fp32 on the device -> (audio_norm (B,T,F), labels (B,T) int64,
    y (B,T,P-2*len(mask))) as the reference hands them to the model."""
    if not (pose_raw.is_cuda and audio_raw.is_cuda) or pose_raw.dtype != torch.float32 or audio_raw.dtype != torch.float32:
      raise TypeError('DevicePreStep takes float32 tensors on the MI355X')
    pose_raw, audio_raw = pose_raw.contiguous(), audio_raw.contiguous()
    B, T, P = pose_raw.shape
    F_ = audio_raw.shape[-1]
    assert P == self.P and audio_raw.shape[:2] == (B, T) and F_ == self.audio_mean.numel()
    s = _vp(torch.cuda.current_stream().cuda_stream)
    labels = torch.empty((B, T), dtype=torch.int64, device=self.dev)
    check(lib().ms_kmeans_labels(_p(pose_raw), _p(self.keep), _p(self.centers), _p(labels), B, T, P, self.PK, self.M, self.feats, s),
          'ms_kmeans_labels')
    y = torch.empty((B, T, self.PK), dtype=torch.float32, device=self.dev)
    check(lib().ms_znorm_select(_p(pose_raw), _p(self.keep), _p(self.pose_mean), _p(self.pose_inv), _p(y), B * T, P, self.PK, s),
          'ms_znorm_select')
    a = torch.empty_like(audio_raw)
    check(lib().ms_znorm_select(_p(audio_raw), None, _p(self.audio_mean), _p(self.audio_inv), _p(a), B * T, F_, F_, s),
          'ms_znorm_select')
    return a, labels, y
