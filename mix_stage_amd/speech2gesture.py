"""Speech2Gesture_D -- the 1-D PatchGAN pose discriminator (reference: src/model/speech2gesture.py:41-74)
on the HIP kernels.  Same constructor, forward(x) -> (scores, []), attribute names and state_dict keys."""
import torch
import torch.nn as nn

from . import ops, ops16
from .layers import ConvNormRelu, bare_conv


class Speech2Gesture_D(nn.Module):
  '''
  input_shape:  (N, time, pose_feats)
  output_shape: (N, *, 1) ## discriminator scores
  '''

  def __init__(self, in_channels=104, out_channels=64, n_downsampling=2, p=0, groups=1, **kwargs):
    super(Speech2Gesture_D, self).__init__()
    self.conv1 = nn.Sequential(torch.nn.Conv1d(in_channels * groups, out_channels * groups, 4, 2, padding=1,
                                               groups=groups),
                               torch.nn.LeakyReLU(negative_slope=0.2))
    self.conv2 = nn.ModuleList([])
    for n in range(1, n_downsampling):
      ch_mul = min(2 ** n, 8)
      self.conv2.append(ConvNormRelu(out_channels, out_channels * ch_mul, type='1d', downsample=True, leaky=True,
                                     p=p, groups=groups))
    self.conv2 = nn.Sequential(*self.conv2)
    ch_mul_new = min(2 ** n_downsampling, 8)
    self.conv3 = ConvNormRelu(out_channels * ch_mul, out_channels * ch_mul_new, type='1d', leaky=True, kernel_size=4,
                              stride=1, p=p, groups=groups)
    out_shape = 1 if 'out_shape' not in kwargs else kwargs['out_shape']
    self.logits = nn.Conv1d(out_channels * ch_mul_new * groups, out_shape * groups, kernel_size=4, stride=1,
                            groups=groups)

  def forward_channel_major(self, x):
    """x: (N, pose_feats, time), e.g. straight from ops.velocity_cm (skips the transpose of S2G:67)."""
    x = bare_conv(self.conv1[0], x, lrelu_slope=self.conv1[1].negative_slope)
    for m in self.conv2:
      x = m(x)
    x = self.conv3(x)
    x = bare_conv(self.logits, x, out_f32=True)                 # scores are fp32 in every mode
    return x.transpose(-1, -2).squeeze(dim=-1), []

  def pair_supported(self, x):
    """Can forward_pair run on x (2B, pose_feats, time) -- every block's kernels implement MS_DT_STAT_PAIR (ms_stat_pair_ok) for this
    batch, train-mode BatchNorm with local statistics, the in-launch meetings available?  Cached per input shape and tuning epoch."""
    from ._lib import MS_BARE, MS_BN_TRAIN, MS_DT_OUT_F32, MS_LRELU
    if not self.training or ops.bn_sync_active() or not ops16.in_launch_meetings():
      return False
    # a caller that hooked a block expects the reference's two calls per D-step, each on a batch of B
    if any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None) for m in self.modules()):
      return False
    dt = getattr(self, '_ms_dt', 0)
    key = (tuple(x.shape), dt, ops.lib().ms_tuning_epoch())
    cache = self.__dict__.setdefault('_pair_ok', {})
    if key not in cache:
      ok, W = x.shape[0] % 2 == 0 and x.dim() == 3, x.shape[-1]
      blocks = [(self.conv1[0], MS_LRELU, None)] + [(m.conv, MS_BN_TRAIN, m) for m in self.conv2] + \
               [(self.conv3.conv, MS_BN_TRAIN, self.conv3), (self.logits, MS_BARE, None)]
      for conv, mode, mod in blocks:
        if not ok:
          break
        if mod is not None and (mod._p or not mod.norm.track_running_stats):
          ok = False
          break
        geom = mod._geometry() if mod is not None else ops.ConvGeom(1, conv.groups, conv.kernel_size, conv.stride, conv.padding)
        if W + 2 * geom.PW < geom.KW:
          ok = False
          break
        flags = (dt | (MS_DT_OUT_F32 if conv is self.logits else 0)) if dt else 0
        ok, W = ops.stat_pair_ok(geom, x.shape[0], conv.weight.shape[1], W, conv.weight.shape[0] // conv.groups, mode, flags)
      cache[key] = bool(ok)
    return cache[key]

  def forward_pair(self, x, split=True):
    """The discriminator on TWO inputs side by side, x = cat([first, second]) channel-major (2B, pose_feats, time; cb8 in the 16-bit
    modes) -> (scores of the first, scores of the second).  Equal to forward_channel_major(first) followed by
    forward_channel_major(second) (gan.py:120,126): BatchNorm statistics per half, the running statistics moved twice in that
    order -- in half the launches.  Call only where pair_supported(x) holds."""
    with ops.stat_pair():
      s = self.forward_channel_major(x)[0]
    return ops.split_halves(s) if split else s              # (split=False: the scores of both, (2B, *), first half first)

  def forward(self, x):
    dt = getattr(self, '_ms_dt', 0)
    if dt:
      return self.forward_channel_major(ops16.btc_to_cb8(x, dt))
    return self.forward_channel_major(ops.to_channel_major(x))
