"""GAN wrapper (reference: src/model/gan.py:18-164) on the HIP kernels: D-step / G-step selection,
pose velocity, GAN + pose losses.  Same constructor (kwargs['input_modalities'] required),
forward(x_audio, y_pose, **kwargs) -> (fake_pose, losses, {'W': W}), flags G_flag / fake_flag / D_prob."""
import contextlib
import os

import torch
import torch.nn as nn

from . import ops, ops16


class ConstantLambdaScheduler:
  """Stand-in for pycasper.torchUtils.LambdaScheduler (not vendored, unpinned; gan.py:30-33,103): returns the
  initial [lambda_D, lambda_gan] on every step().  Inject another object with .step() to change it."""

  def __init__(self, lmbdas, **kwargs):
    self.lmbdas = list(lmbdas)

  def step(self):
    return list(self.lmbdas)

  def state(self):
    return ()

  def set_state(self, state):
    pass


class IncrementalLambdaScheduler:
  """A RAMPING schedule with the reference's constructor arguments (gan.py:30-33: kind='incremental', max_interval=300,
  max_lambda=2).  pycasper's own rule is not available (PARITY UNPINNED: the package is neither vendored nor version-pinned),
  so this is an inferred reading of those arguments, provided so that a non-constant schedule can be trained and captured:
  every `max_interval` calls of step() each lambda grows by its initial value, up to max_lambda times the initial value --
  step() returns the current list.  Pass an instance as GAN(..., lambda_scheduler=...); the default stays constant (what the
  golden vectors were generated with)."""

  def __init__(self, lmbdas, kind='incremental', max_interval=300, max_lambda=2, **kwargs):
    if kind != 'incremental':
      raise NotImplementedError('lambda schedule kind %r' % (kind,))
    self.initial = [float(v) for v in lmbdas]
    self.max_interval, self.max_lambda = int(max_interval), float(max_lambda)
    self.count = 0

  def step(self):
    k = 1 + self.count // self.max_interval
    self.count += 1
    return [min(v * k, v * self.max_lambda) if v >= 0 else max(v * k, v * self.max_lambda) for v in self.initial]

  def state(self):
    return (self.count,)

  def set_state(self, state):
    self.count, = state


_SUPPORTED = {'L1Loss': 'l1_mean', 'MSELoss': 'l2_mean'}      # criterion -> fused mean-reduced kernel (gan.py:40,64-75)


class GAN(nn.Module):
  def __init__(self, G, D, dg_iter_ratio=1, lambda_D=1, lambda_gan=1, lr=0.0001, criterion='MSELoss', optim='Adam',
               joint=False, update_D_prob_flag=True, no_grad=True, **kwargs):
    super(GAN, self).__init__()
    self.G = G
    self.D = D
    self.D_prob = dg_iter_ratio / (dg_iter_ratio + 1)
    self.lambda_D = lambda_D
    self.lambda_gan = lambda_gan
    self.lambda_scheduler = kwargs.get('lambda_scheduler') or ConstantLambdaScheduler(
        [self.lambda_D, self.lambda_gan], kind='incremental', max_interval=300, max_lambda=2)
    # the two loss weights as seen by the kernels: a 2-float device tensor the loss ops read at run time, so that a captured
    # step (HIP graph) follows a schedule that moves between replays.  `_lambda_host_writes`: forward() itself refreshes it
    # (eager use); a trainer that replays captured steps switches that off and writes the values before each replay.
    self._lambda_dev = None
    self._ones_cache = {}
    self._lambda_host_writes = True
    self.G_flag = True
    self.fake_flag = True
    self.lr = lr
    if criterion not in _SUPPORTED:
      raise NotImplementedError('criterion %s: the HIP path implements L1Loss (what src/jobs/mix-stage.py trains '
                                'with) and MSELoss (the constructor default)' % criterion)
    self.criterion_name = criterion
    self._loss = getattr(ops, _SUPPORTED[criterion])
    self.joint = joint
    if joint:
      raise NotImplementedError('joint=True (D sees pose||audio) is not on the Mix-StAGE path')
    self.input_modalities = kwargs['input_modalities']
    self.update_D_prob_flag = update_D_prob_flag
    self.no_grad = no_grad
    # In the G-step the reference also back-propagates into D's weights and then discards those gradients
    # (trainer.py:1104-1107,1140-1142).  Skipping that weight-gradient work changes no result.
    self.skip_D_weight_grads_in_G_step = True
    # D-step: the discriminator sees the fake and the real velocities in one batch with two BatchNorm statistics groups (same
    # results as gan.py:120,126's two passes; MS_PAIR_D=0 / False: the two passes one after the other)
    self.pair_D_passes = os.environ.get('MS_PAIR_D', '1') != '0'
    self._pair_probe = {}

  # API parity helpers -------------------------------------------------------------------------
  def get_velocity(self, x, x_audio=None):
    """(B,T,P) -> (B,T,P) velocity (gan.py:47-52)."""
    return ops.to_time_major(ops.velocity_cm(x))

  def get_real_gt(self, x):
    return torch.ones_like(x)

  def get_fake_gt(self, x):
    return torch.zeros_like(x)

  def get_gan_loss(self, y_cap, y, W=None):
    return self._loss(y_cap, y)

  def get_loss(self, y_cap, y, W=None):
    return self._loss(y_cap, y)

  def estimate_weights(self, x_audio, y_pose, **kwargs):
    # sample weights are 1 (gan.py:77-78): one constant per (batch size, device, dtype), made on the device (a host->device copy
    # would not be capturable in a HIP graph) and reused -- a fill kernel per step otherwise.  Treat it as read-only.
    key = (y_pose.shape[0], y_pose.device, y_pose.dtype)
    W = self._ones_cache.get(key)
    if W is None:
      W = self._ones_cache[key] = torch.ones(y_pose.shape[0], device=y_pose.device, dtype=y_pose.dtype)
    return W, None

  def estimate_weights_loss(self, W):
    return W

  def update_D_prob(self, W):
    pass

  @contextlib.contextmanager
  def _frozen_D(self):
    flags = [(p, p.requires_grad) for p in self.D.parameters()]
    for p, _ in flags:
      p.requires_grad_(False)
    try:
      yield
    finally:
      for p, f in flags:
        p.requires_grad_(f)

  def lambda_device(self, device):
    """The (lambda_D, lambda_gan) device tensor (created on first use)."""
    if self._lambda_dev is None or self._lambda_dev.device != device:
      self._lambda_dev = torch.tensor([float(self.lambda_D), float(self.lambda_gan)], dtype=torch.float32, device=device)
    return self._lambda_dev

  def write_lambdas(self, device, staging=None):
    """Current host values -> device tensor, BY VALUE through a kernel's arguments (ms_write_floats): the values are copied when
    the launch is enqueued, so a host that runs ahead of the device (replayed graphs never synchronise) cannot overwrite them
    before the device has read them -- an asynchronous copy from one pinned staging tensor could (`staging` is ignored)."""
    import ctypes
    dev = self.lambda_device(device)
    # (a launch per step only while the schedule moves: the constant schedule writes once per device tensor)
    now = (float(self.lambda_D), float(self.lambda_gan), dev.data_ptr())
    if getattr(self, '_lambdas_written', None) == now:
      return dev
    vals = (ctypes.c_float * 2)(now[0], now[1])
    with torch.cuda.device(dev.device):
      ops.check(ops.lib().ms_write_floats(ops._ptr(dev), vals, 2, ops._stream()), 'ms_write_floats')
    self._lambdas_written = now
    return dev

  def _score(self, pose):
    """D(get_velocity(pose)) with the velocity produced channel-major for D's first conv."""
    dt = getattr(self.D, '_ms_dt', 0)
    if dt and hasattr(self.D, 'forward_channel_major'):
      return self.D.forward_channel_major(ops16.btc_to_cb8(pose, dt, velocity=True))[0]
    v = ops.velocity_cm(pose)
    if hasattr(self.D, 'forward_channel_major'):
      return self.D.forward_channel_major(v)[0]
    return self.D(ops.to_time_major(v))[0]

  def _score_pair(self, first, second):
    """D(get_velocity(first)) and D(get_velocity(second)) as one (2B, *) score tensor from one pass over both
    (Speech2Gesture_D.forward_pair), or None when the module or this batch has no paired form (hooked modules, global BatchNorm
    statistics, float64 or CPU poses, unequal shapes, a geometry ms_stat_pair_ok declines): the caller then makes the two passes."""
    D = self.D
    if not hasattr(D, 'forward_pair') or first.shape != second.shape or first.dtype != second.dtype or first.dim() != 3:
      return None
    if first.dtype != torch.float32 or not first.is_cuda:
      return None
    B, T, P = first.shape
    probe = self._pair_probe.get((B, T, P, first.device))
    if probe is None:
      probe = self._pair_probe[(B, T, P, first.device)] = torch.empty(2 * B, P, T, device='meta')
    if not D.pair_supported(probe):
      return None
    both = torch.cat([first, second], dim=0)
    dt = getattr(D, '_ms_dt', 0)
    return D.forward_pair(ops16.btc_to_cb8(both, dt, velocity=True) if dt else ops.velocity_cm(both), split=False)

  def forward(self, x_audio, y_pose, **kwargs):
    internal_losses = []
    if 'confidence' in kwargs and not (isinstance(kwargs['confidence'], (int, float)) and kwargs['confidence'] == 1):
      raise NotImplementedError('confidence weighting is not on the Mix-StAGE path (confidence == 1)')
    W, outputs = self.estimate_weights(x_audio, y_pose, **kwargs)      # ones: sample weights are 1 (gan.py:77-78)
    if self.update_D_prob_flag:
      self.update_D_prob(W)

    if self.training:
      self.lambda_D, self.lambda_gan = self.lambda_scheduler.step()
      lam = None
      if y_pose.is_cuda:
        lam = self.write_lambdas(y_pose.device) if self._lambda_host_writes else self.lambda_device(y_pose.device)
      lam_D = lam[0] if lam is not None else self.lambda_D
      lam_gan = lam[1] if lam is not None else self.lambda_gan
      if torch.rand(1).item() < self.D_prob:                           # host RNG draw (gan.py:105)
        ## D-step: G in eval mode under no_grad, D on fake then real velocity (gan.py:106-132)
        self.G.eval()
        with torch.no_grad():
          fake_pose, partial_i_loss, *args = self.G(x_audio, y_pose, **kwargs)
          args = args[0] if len(args) > 0 else {}
        self.G.train(self.training)
        self.fake_flag = True
        pair = self._score_pair(fake_pose.detach(), y_pose) if self.pair_D_passes else None
        if pair is not None:
          # D on the fake and on the real velocities as ONE batch of 2B clips with two BatchNorm statistics groups (include/mixstage.h:
          # MS_DT_STAT_PAIR): the same values as the two passes below, half the launches
          if pair.dtype == torch.float32 and pair.numel() // 2 <= ops.LP_PAIR_MAX:
            # both criterion terms in one launch each way (the same bits as the two calls below)
            fake_D_loss, real_D_loss = ops.lp_mean_pair(pair, (0.0, 1.0), (lam_D, 1.0), squared=self.criterion_name == 'MSELoss')
          else:
            fake_pose_score, real_pose_score = ops.split_halves(pair)
            fake_D_loss = self._loss(fake_pose_score, target=0.0, scale=lam_D)
            real_D_loss = self._loss(real_pose_score, target=1.0)
        else:
          fake_pose_score = self._score(fake_pose.detach())
          fake_D_loss = self._loss(fake_pose_score, target=0.0, scale=lam_D)
          real_pose_score = self._score(y_pose)
          real_D_loss = self._loss(real_pose_score, target=1.0)
        internal_losses.append(real_D_loss)
        internal_losses.append(fake_D_loss)
        internal_losses += partial_i_loss
        self.G_flag = False
      else:
        ## G-step (gan.py:134-152)
        fake_pose, partial_i_loss, *args = self.G(x_audio, y_pose, **kwargs)
        args = args[0] if len(args) > 0 else {}
        if self.no_grad:
          with torch.no_grad():
            fake_pose_score = self._score(fake_pose)
        elif self.skip_D_weight_grads_in_G_step:
          with self._frozen_D():
            fake_pose_score = self._score(fake_pose)
        else:
          fake_pose_score = self._score(fake_pose)
        G_gan_loss = self._loss(fake_pose_score, target=1.0, scale=lam_gan)
        pose_loss = self._loss(fake_pose, y_pose)
        internal_losses.append(pose_loss)
        internal_losses.append(G_gan_loss)
        internal_losses += partial_i_loss
        self.G_flag = True
    else:
      fake_pose, partial_i_loss, *args = self.G(x_audio, y_pose, **kwargs)
      args = args[0] if len(args) > 0 else {}
      pose_loss = self._loss(fake_pose, y_pose)
      internal_losses.append(pose_loss)
      internal_losses.append(torch.tensor(0))
      internal_losses += partial_i_loss
      self.G_flag = True

    args.update(dict(W=W))
    return fake_pose, internal_losses, args
