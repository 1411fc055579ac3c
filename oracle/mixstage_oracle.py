"""CPU ORACLE -- test infrastructure, NOT product code.

Pure-PyTorch (CPU) restatement of the Mix-StAGE audio->pose conditional-mixture
GAN forward/backward path, written from the behaviour of the reference
(chahuja/mix-stage, paths relative to /root/reference/src/model):

  layers.py:32-78    ConvNormRelu        -> Block
  layers.py:80-157   UNet1D
  layers.py:159-199  AudioEncoder
  layers.py:201-240  PoseEncoder
  layers.py:246-289  PoseStyleEncoder
  layers.py:339-373  TextEncoder1D
  layers.py:446-467  ClusterClassify
  layers.py:593-650  Group
  layers.py:652-663  EmbLin
  layers.py:677-696  Curriculum
  joint_late_cluster_soft_style.py:17-209   JointLateClusterSoftStyle4_G
  speech2gesture.py:41-74                   Speech2Gesture_D
  gan.py:18-164                             GAN
  trainer.py:604,1158-1165,1268-1285,1138-1146   train-step contract -> oracle_train_step

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product package (mix_stage_amd) never does.

Pinning: the reference has no tests or golden vectors.  This
restatement is pinned by (i) tests/test_oracle_vs_reference.py, which imports the
reference's own model files in the build container and compares outputs and all
parameter gradients, and (ii) the committed fixtures under tests/golden/ that
tests/golden/make_golden.py generated FROM THE REFERENCE.  Two symbols the
reference takes from the un-vendored, un-pinned `pycasper` package --
`some_grad` and `LambdaScheduler` -- are "parity unpinned": their behaviour is
inferred from the call sites (joint_late_cluster_soft_style.py:198-200,
gan.py:30-33,103).
"""
import contextlib
import math
import zlib

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
def _pow2_factor(n):
  """How many times 2 divides n (layers.py:16-24)."""
  c = 0
  while n > 1 and n % 2 == 0:
    n //= 2
    c += 1
  return c


def default_padding(kernel_size, stride):
  """Padding rule of layers.py:46-55, including the tuple/tuple quirk (always 0)."""
  k_t, s_t = isinstance(kernel_size, tuple), isinstance(stride, tuple)
  if not k_t and s_t:
    return tuple(int((kernel_size - s) / 2) for s in stride)
  if k_t and not s_t:
    return tuple(int((k - stride) / 2) for k in kernel_size)
  if k_t and s_t:
    assert len(kernel_size) == len(stride)
    return tuple(0 for _ in kernel_size)
  return int((kernel_size - stride) / 2)


@contextlib.contextmanager
def some_grad(module):
  """pycasper.torchUtils.some_grad stand-in (PARITY UNPINNED): parameters of `module`
  do not receive gradients inside the context, activations still do."""
  saved = [(p, p.requires_grad) for p in module.parameters()]
  for p, _ in saved:
    p.requires_grad_(False)
  try:
    yield
  finally:
    for p, flag in saved:
      p.requires_grad_(flag)


class ConstantLambdas:
  """pycasper.torchUtils.LambdaScheduler stand-in (PARITY UNPINNED): returns the
  initial lambdas unchanged on every step()."""
  def __init__(self, lambdas, **_):
    self.lambdas = list(lambdas)

  def step(self):
    return list(self.lambdas)


# --------------------------------------------------------------------------
# blocks
# --------------------------------------------------------------------------
class ConvNormRelu(nn.Module):
  """conv(+bias) -> dropout(p) -> BatchNorm -> (Leaky)ReLU   (layers.py:32-78)."""

  def __init__(self, in_channels, out_channels, type='1d', leaky=False, downsample=False,
               kernel_size=None, stride=None, padding=None, p=0, groups=1):
    super().__init__()
    if kernel_size is None and stride is None:
      kernel_size, stride = (4, 2) if downsample else (3, 1)
    if padding is None:
      padding = default_padding(kernel_size, stride)
    conv_cls, norm_cls, drop_cls = {
        '1d': (nn.Conv1d, nn.BatchNorm1d, nn.Dropout),
        '2d': (nn.Conv2d, nn.BatchNorm2d, nn.Dropout2d)}[type]
    self.conv = conv_cls(in_channels * groups, out_channels * groups, kernel_size=kernel_size,
                         stride=stride, padding=padding, groups=groups)
    self.norm = norm_cls(out_channels * groups)
    self.dropout = drop_cls(p=p)
    self.relu = nn.LeakyReLU(0.2) if leaky else nn.ReLU()

  def forward(self, x, **kwargs):
    return self.relu(self.norm(self.dropout(self.conv(x))))


def _stack(specs, **common):
  return nn.ModuleList([ConvNormRelu(ci, co, downsample=ds, **common) for ci, co, ds in specs])


def _run(mods, x):
  for m in mods:
    x = m(x)
  return x


class UNet1D(nn.Module):
  """layers.py:80-157."""

  def __init__(self, input_channels, output_channels, max_depth=5, kernel_size=None, stride=None,
               p=0, groups=1):
    super().__init__()
    kw = dict(type='1d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    c = output_channels
    self.pre_downsampling_conv = _stack([(input_channels, c, False), (c, c, False)], **kw)
    self.conv1 = _stack([(c, c, True)] * max_depth, **kw)
    self.conv2 = _stack([(c, c, False)] * max_depth, **kw)
    self.upconv = nn.Upsample(scale_factor=2, mode='nearest')
    self.max_depth, self.groups = max_depth, groups

  def forward(self, x, return_bottleneck=False):
    T = x.shape[-1]
    assert T / (2 ** (self.max_depth - 1)) >= 1
    assert _pow2_factor(T) >= self.max_depth, 'time axis must be a multiple of 2^max_depth'
    x = _run(self.pre_downsampling_conv, x)
    skips = [x]
    for i, down in enumerate(self.conv1):
      x = down(x)
      if i < self.max_depth - 1:
        skips.append(x)
    bottleneck = x
    for i, up in enumerate(self.conv2):
      x = up(self.upconv(x) + skips[self.max_depth - 1 - i])
    return (x, bottleneck) if return_bottleneck else x


class AudioEncoder(nn.Module):
  """layers.py:159-199: 8 Conv2d blocks over (N,1,T,F), then bilinear resize to (T,1)."""

  def __init__(self, output_feats=64, input_channels=1, kernel_size=None, stride=None, p=0, groups=1):
    super().__init__()
    kw = dict(type='2d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    self.conv = _stack([(input_channels, 64, False), (64, 64, True), (64, 128, False),
                        (128, 128, True), (128, 256, False), (256, 256, True),
                        (256, 256, False)], **kw)
    self.conv.append(ConvNormRelu(256, 256, type='2d', leaky=True, downsample=False,
                                  kernel_size=(3, 8), stride=1, p=p, groups=groups))

  def forward(self, x, time_steps=None):
    if time_steps is None:
      time_steps = x.shape[-2]
    x = _run(self.conv, x)
    x = F.interpolate(x, size=(time_steps, 1), mode='bilinear')
    return x.squeeze(dim=-1)


class _Seq1D(nn.Module):
  """(N,T,C) -> transpose -> chain of 1-D blocks (shared shape of the pose/text encoders)."""
  _specs = ()

  def __init__(self, output_feats=64, input_channels=None, kernel_size=None, stride=None, p=0,
               groups=1, **extra):
    super().__init__()
    kw = dict(type='1d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    self.conv = _stack(self._make_specs(input_channels, **extra), **kw)

  def _body(self, x):
    return _run(self.conv, torch.transpose(x, 1, 2))


def _flat_specs(cin):
  return [(cin, 64, False), (64, 64, False), (64, 128, False), (128, 128, False),
          (128, 256, False), (256, 256, False)]


class PoseEncoder(_Seq1D):
  """layers.py:201-240."""
  _make_specs = staticmethod(_flat_specs)

  def __init__(self, output_feats=64, input_channels=96, **kw):
    super().__init__(output_feats, input_channels, **kw)

  def forward(self, x, time_steps=None):
    return self._body(x).squeeze(dim=-1)


class TextEncoder1D(_Seq1D):
  """layers.py:339-373."""

  _make_specs = staticmethod(_flat_specs)

  def __init__(self, output_feats=64, input_channels=300, **kw):
    super().__init__(output_feats, input_channels, **kw)

  def forward(self, x, time_steps=None, **kwargs):
    return self._body(x).squeeze(dim=-1)


class PoseStyleEncoder(_Seq1D):
  """layers.py:246-289: one k3 block, six k4/s2 blocks (last one -> num_speakers, still with
  BN + LeakyReLU), then mean over the remaining time axis."""

  def __init__(self, output_feats=64, input_channels=96, kernel_size=None, stride=None, p=0,
               groups=1, num_speakers=4):
    super().__init__(output_feats, input_channels, kernel_size=kernel_size, stride=stride, p=p,
                     groups=groups, num_speakers=num_speakers)

  @staticmethod
  def _make_specs(cin, num_speakers=4):
    return [(cin, 64, False), (64, 64, True), (64, 128, True), (128, 128, True),
            (128, 256, True), (256, 256, True), (256, num_speakers, True)]

  def forward(self, x, time_steps=None):
    return self._body(x).mean(-1).squeeze(dim=-1)


class ClusterClassify(nn.Module):
  """layers.py:446-467."""

  def __init__(self, num_clusters=8, kernel_size=None, stride=None, p=0, groups=1, input_channels=256):
    super().__init__()
    kw = dict(type='1d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    self.conv = _stack([(input_channels, 256, False)] + [(256, 256, False)] * 5, **kw)
    self.logits = nn.Conv1d(256 * groups, num_clusters * groups, kernel_size=1, stride=1, groups=groups)

  def forward(self, x, time_steps=None):
    return self.logits(_run(self.conv, x))


def mix_outputs(z, weights, groups):
  """index_select_outputs (joint_late_cluster_soft_style.py:106-115, layers.py:617-626):
  z (B, groups*P, T), weights (B,T,groups) -> (B,T,P) = sum_m w[b,t,m] * z[b, m*P+f, t]."""
  z = z.transpose(2, 1)
  z = z.view(z.shape[0], z.shape[1], groups, -1)
  weights = weights.view(z.shape[0], z.shape[1], z.shape[2])
  return (z * weights.unsqueeze(-1)).sum(dim=-2)


class Group(nn.Module):
  """layers.py:593-650 (only constructed on the path, never called)."""

  def __init__(self, models, groups=1, dim=1):
    super().__init__()
    self.models = nn.ModuleList(models if isinstance(models, list) else [models])
    self.groups, self.dim = groups, dim

  def forward(self, x, labels=None, transpose=True, **kwargs):
    if self.dim == 0:
      self.groups = len(x)
    if isinstance(x, list):
      x = torch.cat(x, dim=self.dim)
    if transpose:
      x = x.transpose(-1, -2)
    for m in self.models:
      x = m(x, **kwargs) if kwargs else m(x)
    if labels is not None:
      return mix_outputs(x, labels, self.groups).transpose(-1, -2)
    ch = int(x.shape[self.dim] / self.groups)
    return list(torch.split(x, ch, dim=self.dim % x.dim()))


class EmbLin(nn.Module):
  """layers.py:652-663."""

  def __init__(self, num_embeddings, embedding_dim):
    super().__init__()
    self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
    self.emb = nn.Embedding(num_embeddings, embedding_dim)

  def forward(self, x, mode='lin'):
    if mode == 'lin':
      return x.matmul(self.emb.weight)
    if mode == 'emb':
      return self.emb(x)


class Curriculum:
  """layers.py:677-696: linear ramp start->end over num_iters *training* calls."""

  def __init__(self, start, end, num_iters):
    self.start, self.end, self.num_iters = start, end, num_iters
    self.iters = 0
    self.diff = (end - start) / num_iters
    self.value = start

  def step(self, flag=True):
    if not flag:
      return self.value
    if self.iters >= self.num_iters:
      return self.end
    before = self.value
    self.value += self.diff
    self.iters += 1
    return before


# --------------------------------------------------------------------------
# generator / discriminator / GAN
# --------------------------------------------------------------------------
class JointLateClusterSoftStyle4_G(nn.Module):
  """joint_late_cluster_soft_style.py:17-209."""

  def __init__(self, time_steps=64, in_channels=256, out_feats=104, p=0, num_clusters=8, cluster=None,
               style_dict={}, style_dim=10, lambda_id=1, train_only=0, softmax=1, argmax=0,
               some_grad_flag=False, **kwargs):
    super().__init__()
    S = len(style_dict)
    self.num_clusters, self.style_dict, self.style_dim = num_clusters, style_dict, style_dim
    self.lambda_id, self.train_only = lambda_id, train_only
    self.softmax, self.argmax, self.some_grad_flag = softmax, argmax, some_grad_flag
    self.cluster = cluster

    self.audio_encoder = AudioEncoder(output_feats=time_steps, p=p)
    text_key = None
    for key in kwargs['shape']:
      if key in ('text/w2v', 'text/bert'):
        text_key = key
    if text_key:
      self.text_encoder = TextEncoder1D(output_feats=time_steps,
                                        input_channels=kwargs['shape'][text_key][-1], p=p)
    else:
      self.text_encoder = TextEncoder1D(output_feats=time_steps, p=p)
    self.pose_encoder = PoseEncoder(output_feats=time_steps, input_channels=out_feats, p=p)
    self.unet = UNet1D(in_channels, in_channels, p=p, groups=1)

    self.pose_style_encoder = PoseStyleEncoder(input_channels=out_feats, p=p, num_speakers=S)
    self.style_emb = EmbLin(S, style_dim)
    lk = dict(type='1d', leaky=True, downsample=False, p=p)
    self.style_dec = nn.Sequential(*[ConvNormRelu(in_channels, in_channels, groups=style_dim, **lk)
                                     for _ in range(2)])
    self.style_dec_gr = Group([self.style_dec], groups=style_dim)

    self.decoder = nn.Sequential(
        ConvNormRelu(style_dim + in_channels, in_channels, groups=num_clusters, **lk),
        *[ConvNormRelu(in_channels, in_channels, groups=num_clusters, **lk) for _ in range(3)])
    self.concat_encoder = nn.Sequential(ConvNormRelu(512, 256, **lk))
    self.logits = nn.Conv1d(in_channels * num_clusters, out_feats * num_clusters, kernel_size=1,
                            stride=1, groups=num_clusters)
    self.classify_cluster = ClusterClassify(num_clusters=num_clusters, groups=1,
                                            input_channels=style_dim + in_channels)
    self.classify_loss = nn.CrossEntropyLoss()
    self.eye = nn.Parameter(torch.eye(num_clusters, num_clusters), requires_grad=False)
    self.smoothen = ConvNormRelu(out_feats, out_feats, **lk)

    self.thresh = Curriculum(0, 1, 1000)
    self.labels_cap_soft = None

  def index_select_outputs(self, x, labels, groups):
    return mix_outputs(x, labels, groups)

  def forward(self, x, y, time_steps=None, **kwargs):
    labels, x = x[-1], list(x[:-1])
    # host RNG draw happens on every call, training or not (line 127)
    use_pose_branch = torch.rand(1).item() > self.thresh.step(self.training) and self.training
    if use_pose_branch:
      x = self.pose_encoder(y, time_steps)
    else:
      for i, modality in enumerate(kwargs['input_modalities']):
        kind = modality.split('/')[0]
        if kind == 'text':
          x[i] = self.text_encoder(x[i], time_steps)
        if kind == 'audio':
          if x[i].dim() == 3:
            x[i] = x[i].unsqueeze(dim=1)
          x[i] = self.audio_encoder(x[i], time_steps)
      fused = torch.cat(tuple(x), dim=1)
      x = self.concat_encoder(fused) if len(x) >= 2 else fused

    x = self.unet(x).transpose(2, 1)                               # (B,T,256)

    style = kwargs['style']
    use_pse = (not kwargs['sample_flag']) and (kwargs['description'] == 'train' or not self.train_only)
    if use_pse:
      mode = 'lin'
      score = self.pose_style_encoder(y)                           # (B,S)
      id_in = F.cross_entropy(score, style[:, 0])
      score = score.unsqueeze(1).expand(score.shape[0], x.shape[1], score.shape[-1])
      if self.softmax:
        pose_style = torch.softmax(score, dim=-1)
        if self.argmax:
          pose_style = torch.argmax(pose_style, dim=-1)
          mode = 'emb'
      else:
        pose_style = score
    else:
      pose_style = style
      mode = 'emb' if style.dim() == 2 else 'lin'
      id_in = torch.zeros(1)[0]
    self.pose_style_ids = pose_style if mode == 'emb' else None     # oracle-only probe
    emb = self.style_emb(pose_style, mode=mode)
    if x.shape[1] != emb.shape[1]:
      emb = emb.view(x.shape[0], -1, emb.shape[-1])
    x = torch.cat([x, emb], dim=-1).transpose(2, 1)                # (B,266,T)

    labels_score = self.classify_cluster(x).transpose(2, 1)        # (B,T,M)
    losses = [self.classify_loss(labels_score.reshape(-1, labels_score.shape[-1]), labels.reshape(-1))]
    soft = F.softmax(labels_score, dim=-1)
    self.labels_cap_soft = soft

    x = torch.cat([x] * self.num_clusters, dim=1)
    x = self.logits(self.decoder(x))
    x = mix_outputs(x, soft, self.num_clusters)                    # (B,T,P)

    if use_pse:
      if self.some_grad_flag:
        with some_grad(self.pose_style_encoder):
          score_out = self.pose_style_encoder(x)
      else:
        score_out = self.pose_style_encoder(x)
      id_out = F.cross_entropy(score_out, style[:, 0])
    else:
      id_out = torch.zeros(1)[0]
    losses.append(id_in * self.lambda_id)
    losses.append(id_out * self.lambda_id)
    return x, losses


class Speech2Gesture_D(nn.Module):
  """speech2gesture.py:41-74: 1-D PatchGAN over pose velocity."""

  def __init__(self, in_channels=104, out_channels=64, n_downsampling=2, p=0, groups=1, **kwargs):
    super().__init__()
    self.conv1 = nn.Sequential(nn.Conv1d(in_channels * groups, out_channels * groups, 4, 2, padding=1,
                                         groups=groups), nn.LeakyReLU(0.2))
    mids, mul = [], None
    for n in range(1, n_downsampling):
      mul = min(2 ** n, 8)
      mids.append(ConvNormRelu(out_channels, out_channels * mul, type='1d', downsample=True,
                               leaky=True, p=p, groups=groups))
    self.conv2 = nn.Sequential(*mids)
    mul_new = min(2 ** n_downsampling, 8)
    self.conv3 = ConvNormRelu(out_channels * mul, out_channels * mul_new, type='1d', leaky=True,
                              kernel_size=4, stride=1, p=p, groups=groups)
    out_shape = kwargs.get('out_shape', 1)
    self.logits = nn.Conv1d(out_channels * mul_new * groups, out_shape * groups, kernel_size=4,
                            stride=1, groups=groups)

  def forward(self, x):
    x = self.logits(self.conv3(self.conv2(self.conv1(x.transpose(-1, -2)))))
    return x.transpose(-1, -2).squeeze(dim=-1), []


class GAN(nn.Module):
  """gan.py:18-164."""

  def __init__(self, G, D, dg_iter_ratio=1, lambda_D=1, lambda_gan=1, lr=0.0001, criterion='MSELoss',
               optim='Adam', joint=False, update_D_prob_flag=True, no_grad=True, **kwargs):
    super().__init__()
    self.G, self.D = G, D
    self.D_prob = dg_iter_ratio / (dg_iter_ratio + 1)
    self.lambda_D, self.lambda_gan = lambda_D, lambda_gan
    self.lambda_scheduler = kwargs.get('lambda_scheduler') or ConstantLambdas([lambda_D, lambda_gan])
    self.G_flag = True
    self.fake_flag = True
    self.lr = lr
    self.criterion = getattr(nn, criterion)(reduction='none')
    self.joint = joint
    self.input_modalities = kwargs['input_modalities']
    self.update_D_prob_flag = update_D_prob_flag
    self.no_grad = no_grad

  def get_velocity(self, x, x_audio):
    v = torch.cat([torch.zeros_like(x[..., 0:1, :]), x[..., 1:, :] - x[..., :-1, :]], dim=-2)
    if self.joint:
      return torch.cat([v, torch.cat(x_audio[:len(self.input_modalities)], dim=-1)], dim=-1)
    return v

  def _wmean(self, loss, W):
    W = W.view([W.shape[0]] + [1] * (loss.dim() - 1))
    return (W.expand_as(loss) * loss).mean()

  def get_gan_loss(self, y_cap, y, W):
    return self._wmean(self.criterion(y_cap, y), W)

  get_loss = get_gan_loss

  def estimate_weights(self, x_audio, y_pose, **kwargs):
    return torch.ones(y_pose.shape[0]).to(y_pose.device), None

  def forward(self, x_audio, y_pose, **kwargs):
    confidence = kwargs.get('confidence', 1)
    W, _ = self.estimate_weights(x_audio, y_pose, **kwargs)
    losses, extra = [], {}
    if self.training:
      self.lambda_D, self.lambda_gan = self.lambda_scheduler.step()
      if torch.rand(1).item() < self.D_prob:                       # D-step (gan.py:105-132)
        self.G.eval()
        with torch.no_grad():
          fake, partial = self.G(x_audio, y_pose, **kwargs)[:2]
        self.G.train(self.training)
        real_v = self.get_velocity(y_pose, x_audio)
        fake_v = self.get_velocity(fake, x_audio)
        self.fake_flag = True
        fake_score, _ = self.D(fake_v.detach())
        fake_loss = self.lambda_D * self.get_gan_loss(fake_score, torch.zeros_like(fake_score),
                                                       torch.ones_like(1 / W))
        real_score, _ = self.D(real_v)
        real_loss = self.get_gan_loss(real_score, torch.ones_like(real_score), torch.ones_like(W))
        losses += [real_loss, fake_loss] + list(partial)
        self.G_flag = False
      else:                                                        # G-step (gan.py:134-152)
        fake, partial = self.G(x_audio, y_pose, **kwargs)[:2]
        fake_v = self.get_velocity(fake, x_audio)
        if self.no_grad:
          with torch.no_grad():
            fake_score, _ = self.D(fake_v)
        else:
          fake_score, _ = self.D(fake_v)
        gan_loss = self.lambda_gan * self.get_gan_loss(fake_score, torch.ones_like(fake_score), 1 / W)
        pose_loss = self.get_loss(fake * confidence, y_pose * confidence, 1 / W)
        losses += [pose_loss, gan_loss] + list(partial)
        self.G_flag = True
    else:
      fake, partial = self.G(x_audio, y_pose, **kwargs)[:2]
      pose_loss = self.get_loss(fake * confidence, y_pose * confidence, torch.ones_like(W))
      losses += [pose_loss, torch.tensor(0)] + list(partial)
      self.G_flag = True
    extra.update(dict(W=W))
    return fake, losses, extra


# --------------------------------------------------------------------------
# deterministic weights / synthetic inputs / the train-step contract
# --------------------------------------------------------------------------
def deterministic_state(state_dict, dtype=torch.float32):
  """Name-keyed deterministic fill: independent of construction
  order and of the global RNG.  Returns a new dict with the same keys/shapes."""
  out = {}
  for k, v in state_dict.items():
    g = torch.Generator().manual_seed(zlib.crc32(k.encode()))
    leaf = k.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
      t = torch.zeros_like(v)
    elif k.endswith('eye'):
      t = v.clone()
    elif leaf == 'running_mean':
      t = torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1
    elif leaf == 'running_var':
      t = 0.5 + torch.rand(v.shape, generator=g, dtype=torch.float64)
    elif '.norm.' in k and leaf == 'weight':
      t = 0.5 + torch.rand(v.shape, generator=g, dtype=torch.float64)
    elif leaf == 'bias':
      t = torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1
    else:  # conv / embedding weights
      fan_in = max(1, int(v[0].numel())) if v.dim() > 1 else 1
      t = torch.randn(v.shape, generator=g, dtype=torch.float64) / math.sqrt(fan_in)
    out[k] = t.to(v.dtype if not v.is_floating_point() else dtype)
  return out


def synthetic_batch(B, T=64, F_=128, P=104, M=8, S=8, seed=1234, dtype=torch.float32):
  """Synthetic clip batch."""
  g = torch.Generator().manual_seed(seed)
  audio = torch.randn(B, T, F_, generator=g, dtype=torch.float64).to(dtype)
  pose = torch.randn(B, T, P, generator=g, dtype=torch.float64).to(dtype)
  labels = torch.randint(0, M, (B, T), generator=g, dtype=torch.int64)
  style = (torch.arange(B, dtype=torch.int64) % S).unsqueeze(1).expand(B, T).contiguous()
  return audio, pose, labels, style


def build_gan(M=8, S=8, T=64, P=104, dtype=torch.float32, lambda_id=0.1, no_grad=0, **gkw):
  """The configuration the job scripts train (src/jobs/mix-stage.py:3)."""
  G = JointLateClusterSoftStyle4_G(time_steps=T, out_feats=P, num_clusters=M,
                                   style_dict={i: i for i in range(S)}, style_dim=10,
                                   lambda_id=lambda_id, argmax=1, some_grad_flag=1, train_only=1,
                                   shape={}, **gkw)
  D = Speech2Gesture_D(in_channels=P)
  model = GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'],
              update_D_prob_flag=0, no_grad=no_grad)
  model.load_state_dict(deterministic_state(model.state_dict()))
  model.to(dtype)
  model.G.thresh.value, model.G.thresh.iters = 1, 10 ** 9         # pin the audio branch
  return model


def model_kwargs(style, T=64):
  return dict(input_modalities=['audio/log_mel_400'], desc='train', sample_flag=0,
              description='train', style=style, time_steps=T)


def oracle_train_step(model, optim_G, optim_D, audio, pose, labels, style, step_kind, T=64):
  """One trainer step (trainer.py:604 zero_grad, :1158-1165 forward, :1268-1285 loss = sum,
  :1138-1146 backward + clip_grad_norm_(.,1) + Adam on G or D).  `step_kind` in {'G','D'}
  pins gan.py:105's coin flip."""
  model.train()
  # The reference pins torch==1.5.0 (requirements.txt:13), whose zero_grad() ZEROES existing gradients instead of dropping
  # them: a parameter that has received a gradient once keeps being updated by Adam (zero gradient, decaying momentum) in
  # steps where it is unused, and a parameter that never had one is skipped.  set_to_none=False reproduces that.
  model.zero_grad(set_to_none=False)
  optim_G.zero_grad(set_to_none=False)
  optim_D.zero_grad(set_to_none=False)
  model.D_prob = 1.1 if step_kind == 'D' else -1.0
  fake, losses, _ = model([audio, labels], pose, **model_kwargs(style, T))
  loss = 0
  for l in losses:
    loss = loss + l
  loss.backward()
  if model.G_flag:
    gn = torch.nn.utils.clip_grad_norm_(model.G.parameters(), 1)
    optim_G.step()
  else:
    gn = torch.nn.utils.clip_grad_norm_(model.D.parameters(), 1)
    optim_D.step()
  return fake.detach(), [float(l) for l in losses], float(gn)
