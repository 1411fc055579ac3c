"""Host-side mirror of the reference's building blocks (src/model/layers.py) on the HIP kernels.

Same class names, constructor signatures, forward() signatures, sub-module attribute names and
state_dict keys as the reference, so checkpoints and callers carry over unchanged:

  ConvNormRelu      layers.py:32-78      UNet1D        layers.py:80-157
  AudioEncoder      layers.py:159-199    PoseEncoder   layers.py:201-240
  PoseStyleEncoder  layers.py:246-289    TextEncoder1D layers.py:339-373
  ClusterClassify   layers.py:446-467    Group         layers.py:593-650
  EmbLin            layers.py:652-663    Curriculum    layers.py:677-696

`self.conv` / `self.norm` are torch.nn containers for the parameters and running statistics only
(same default initialisation and RNG consumption as the reference); their forward() is never called --
the arithmetic runs in libmixstage_hip.so (ops.conv_block).
"""
import torch
import torch.nn as nn

from . import ops, ops16
from ._lib import MS_BARE, MS_BN_EVAL, MS_BN_TRAIN, MS_IN_BCAST, MS_IN_PLAIN, MS_IN_UP2ADD, MS_LRELU


def set_compute_dtype(module, dtype):
  """Arithmetic mode of every conv block under `module`: 'fp32' (default: exact fp32 matrix products, the parity path),
  'bf16' or 'fp16' (16-bit operands and activations, fp32 accumulate / BatchNorm statistics / parameters; what casting the
  reference model selects, trainer.py:138).  Parameters and buffers stay fp32 tensors; the modules keep accepting and
  returning fp32 tensors at their public boundaries."""
  code = 0 if dtype in (None, 'fp32', 'f32', 'float32', torch.float32) else (
      ops16.MS_DT[dtype] if isinstance(dtype, torch.dtype) else ops16.NAME_DT[dtype])
  for m in module.modules():
    m._ms_dt = code
  return module


def set_inference_folding(module, on=True):
  """16-bit modes, eval only: fold every eval-mode BatchNorm into the prepared conv weights and bias (valid while the
  running statistics do not change: sampling / style transfer)."""
  for m in module.modules():
    if isinstance(m, ConvNormRelu):
      m._bn_folded = bool(on)
  return module


def compute_dtype(module):
  return getattr(module, '_ms_dt', 0)


# set to a list while a training step is being graph-captured: the ConvNormRelu blocks that ran with batch
# statistics are recorded so graph replays can keep their num_batches_tracked counts right (train_step.py)
_train_tape = None


def set_train_tape(tape):
  global _train_tape
  _train_tape = tape


def num_powers_of_two(x):
  n = 0
  while x > 1 and x % 2 == 0:
    x //= 2
    n += 1
  return n


def next_multiple_power_of_two(x, power=5):
  have = num_powers_of_two(x)
  return x * (2 ** (power - have)) if have < power else x


def _default_padding(kernel_size, stride):
  # layers.py:46-55 (tuple/tuple is (ks-ks)/2 = 0 in the reference: kept bug-compatible)
  kt, st = isinstance(kernel_size, tuple), isinstance(stride, tuple)
  if not kt and st:
    return tuple(int((kernel_size - s) / 2) for s in stride)
  if kt and not st:
    return tuple(int((k - stride) / 2) for k in kernel_size)
  if kt and st:
    assert len(kernel_size) == len(stride), \
        'dims in kernel_size are {} and stride are {}. They must be the same'.format(len(kernel_size), len(stride))
    return tuple(0 for _ in kernel_size)
  return int((kernel_size - stride) / 2)


def bare_conv(conv, x, lrelu_slope=None, out_f32=False):
  """An nn.Conv1d/2d container evaluated on the HIP kernels (JL:83 logits, layers.py:459, S2G:50-51,63).
  16-bit modes: x cb8 (a plain fp32 tensor is converted), result cb8 -- or plain fp32 with out_f32 (score outputs)."""
  geom = getattr(conv, '_ms_geom', None)
  if geom is None:
    nd = 1 if isinstance(conv, nn.Conv1d) else 2
    geom = ops.ConvGeom(nd, conv.groups, conv.kernel_size, conv.stride, conv.padding,
                        slope=0.0 if lrelu_slope is None else lrelu_slope)
    conv._ms_geom = geom
  mode = MS_BARE if lrelu_slope is None else MS_LRELU
  dt = getattr(conv, '_ms_dt', 0)
  if dt:
    was_plain = not ops16.is_cb8(x)
    if was_plain:
      x = ops16.to_cb8(x, dt)
    y = ops16.conv_block16(x, conv.weight, conv.bias, geom, mode, out_f32=out_f32)
    return ops16.from_cb8(y, conv.weight.shape[0]) if (was_plain and not out_f32) else y
  return ops.conv_block(x, conv.weight, conv.bias, geom, mode)


class ConvNormRelu(nn.Module):
  def __init__(self, in_channels, out_channels, type='1d', leaky=False, downsample=False, kernel_size=None,
               stride=None, padding=None, p=0, groups=1):
    super(ConvNormRelu, self).__init__()
    if kernel_size is None and stride is None:
      kernel_size, stride = (4, 2) if downsample else (3, 1)
    if padding is None:
      padding = _default_padding(kernel_size, stride)
    in_channels, out_channels = in_channels * groups, out_channels * groups
    if type == '1d':
      self.conv = nn.Conv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                            stride=stride, padding=padding, groups=groups)
      self.norm = nn.BatchNorm1d(out_channels)
      self.dropout = nn.Dropout(p=p)
    elif type == '2d':
      self.conv = nn.Conv2d(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size,
                            stride=stride, padding=padding, groups=groups)
      self.norm = nn.BatchNorm2d(out_channels)
      self.dropout = nn.Dropout2d(p=p)
    self.relu = nn.LeakyReLU(negative_slope=0.2) if leaky else nn.ReLU()
    self._nd = 1 if type == '1d' else 2
    self._slope = 0.2 if leaky else 0.0
    self._p = p
    self._geom = None
    # nn.BatchNorm's num_batches_tracked is kept as a host-side pending count and folded into the buffer
    # when the state is read: no per-layer device increment inside the (graph-captured) step.
    self._pending_batches = 0
    self._register_state_dict_hook(ConvNormRelu._flush_hook)
    self._register_load_state_dict_pre_hook(self._reset_pending)

  @staticmethod
  def _flush_hook(module, state_dict, prefix, local_metadata):
    if module._pending_batches and module.norm.num_batches_tracked is not None:
      module.norm.num_batches_tracked += module._pending_batches
      module._pending_batches = 0
      state_dict[prefix + 'norm.num_batches_tracked'] = module.norm.num_batches_tracked.detach()

  def _reset_pending(self, *args, **kwargs):
    self._pending_batches = 0

  def _geometry(self):
    if self._geom is None:
      c, n = self.conv, self.norm
      self._geom = ops.ConvGeom(self._nd, c.groups, c.kernel_size, c.stride, c.padding, slope=self._slope,
                                eps=n.eps, momentum=0.1 if n.momentum is None else n.momentum)
    return self._geom

  def _note_train_pass(self):
    """Bookkeeping of one forward pass with batch statistics (nn.BatchNorm's num_batches_tracked, see __init__)."""
    self._pending_batches += 1
    if _train_tape is not None:
      _train_tape.append(self)

  def _run(self, x, x2=None, in_mode=MS_IN_PLAIN, out_f32=False, chain_prev=False, link=None):
    if self._p and self.training:
      raise NotImplementedError('dropout p>0 is not on the Mix-StAGE path (p=0 everywhere, JL:26)')
    n = self.norm
    if self.training and n.track_running_stats:
      mode = MS_BN_TRAIN
      self._note_train_pass()
      if ops.stat_pair_active():
        self._note_train_pass()          # two passes side by side in this batch (ops.stat_pair): two tracked batches
    else:
      mode = MS_BN_EVAL
    dt = getattr(self, '_ms_dt', 0)
    if dt:
      # 16-bit mode: cb8 in, cb8 out; a plain fp32 caller (the public module boundary) is converted both ways
      was_plain = not ops16.is_cb8(x)
      if was_plain:
        x = ops16.to_cb8(x, dt)
        x2 = ops16.to_cb8(x2, dt) if x2 is not None else None
      if mode == MS_BN_TRAIN and ops.bn_sync_active():
        # data parallel with bn_sync='global' in the 16-bit modes: the conv leaves its fp32 accumulators as a plain fp32 tensor,
        # BatchNorm over the batch of ALL ranks runs on that (fp32 statistics, two small collectives, as in the fp32 mode), and
        # the activation goes back to 16 bits for the next block.  Not the one-launch block: a meeting across ranks inside a
        # launch would need device-initiated communication.
        g = self._geometry()
        y_raw = ops16.conv_block16(x, self.conv.weight, self.conv.bias, g, MS_BARE, x2=x2, in_mode=in_mode, out_f32=True)
        y = ops.sync_bn_act(y_raw, n.weight, n.bias, n.running_mean, n.running_var, g.slope, g.eps, g.momentum)
        return y if (was_plain or out_f32) else ops16.to_cb8(y, dt)
      y = ops16.conv_block16(x, self.conv.weight, self.conv.bias, self._geometry(), mode, n.weight, n.bias, n.running_mean,
                             n.running_var, x2=x2, in_mode=in_mode, out_f32=out_f32,
                             bn_folded=getattr(self, '_bn_folded', False))
      return ops16.from_cb8(y, self.conv.weight.shape[0]) if (was_plain and not out_f32) else y
    if mode == MS_BN_TRAIN and ops.bn_sync_active():
      # data parallel with bn_sync='global': bare conv, then BatchNorm over the batch of ALL ranks (two small collectives)
      g = self._geometry()
      y_raw = ops.conv_block(x, self.conv.weight, self.conv.bias, g, MS_BARE, x2=x2, in_mode=in_mode)
      return ops.sync_bn_act(y_raw, n.weight, n.bias, n.running_mean, n.running_var, g.slope, g.eps, g.momentum)
    return ops.conv_block(x, self.conv.weight, self.conv.bias, self._geometry(), mode, n.weight, n.bias,
                          n.running_mean, n.running_var, x2=x2, in_mode=in_mode, chain_prev=chain_prev and mode == MS_BN_TRAIN,
                          link=link)

  def forward(self, x, **kwargs):
    # `_residual` / `_broadcast` / `_out_f32` select the fused forms below (package-internal; the reference's
    # forward(x, **kwargs) ignores its kwargs)
    residual = kwargs.get('_residual')
    out_f32 = bool(kwargs.get('_out_f32'))
    if residual is not None:
      return self._run(x, x2=residual, in_mode=MS_IN_UP2ADD, out_f32=out_f32, link=kwargs.get('_ms_link'))
    if kwargs.get('_broadcast'):
      return self._run(x, in_mode=MS_IN_BCAST, out_f32=out_f32)
    # `_ms_chain`: set by the 1-D stacks of this file for every block but the first -- x is the previous block's output and feeds
    # nothing else, so the backward pass may fuse that block's BatchNorm backward into this block's data gradient (ops.conv_block)
    return self._run(x, out_f32=out_f32, chain_prev=bool(kwargs.get('_ms_chain')), link=kwargs.get('_ms_link'))

  def forward_upsample_add(self, a, residual, _ms_link=None):
    """== self(upsample_nearest2(a) + residual) without materialising the sum (layers.py:151)."""
    return self(a, _residual=residual, _ms_link=_ms_link)

  def forward_broadcast(self, x):
    """== self(torch.cat([x]*groups, dim=1)) without the replication (JL:190)."""
    return self(x, _broadcast=True)


def _hooked(*mods):
  """Does anything outside this package get to see these modules' inputs, outputs or gradients?  Module hooks of any kind
  (forward, forward-pre, full / legacy backward, backward-pre) or process-wide module hooks.  A hooked block keeps its
  BatchNorm backward in its own launch: the fused form (ops.conv_block, chain_prev) hands the producer dy_raw in place of the
  gradient of its output, which is only right while the output feeds exactly one consumer and nobody looks at its gradient."""
  import torch.nn.modules.module as M
  if (M._global_forward_hooks or M._global_forward_pre_hooks or M._global_backward_hooks or
      getattr(M, '_global_backward_pre_hooks', None) or getattr(M, '_global_forward_hooks_always_called', None)):
    return True
  for m in mods:
    if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or getattr(m, '_backward_pre_hooks', None):
      return True
  return False


def _chain_ok(stack, i):
  """Block i of a 1-D stack may fuse block i - 1's BatchNorm backward into its data gradient: i - 1's output is a local of the
  stack's forward() that feeds block i only, unless a hook hands it to somebody else."""
  return i > 0 and not _hooked(stack[i - 1], stack[i])


class UNet1D(nn.Module):
  def __init__(self, input_channels, output_channels, max_depth=5, kernel_size=None, stride=None, p=0, groups=1):
    super(UNet1D, self).__init__()
    self.pre_downsampling_conv = nn.ModuleList([])
    self.conv1 = nn.ModuleList([])
    self.conv2 = nn.ModuleList([])
    self.upconv = nn.Upsample(scale_factor=2, mode='nearest')
    self.max_depth = max_depth
    self.groups = groups
    common = dict(type='1d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    self.pre_downsampling_conv.append(ConvNormRelu(input_channels, output_channels, downsample=False, **common))
    self.pre_downsampling_conv.append(ConvNormRelu(output_channels, output_channels, downsample=False, **common))
    for _ in range(self.max_depth):
      self.conv1.append(ConvNormRelu(output_channels, output_channels, downsample=True, **common))
    for _ in range(self.max_depth):
      self.conv2.append(ConvNormRelu(output_channels, output_channels, downsample=False, **common))

  def forward(self, x, return_bottleneck=False):
    input_size = x.shape[2] if ops16.is_cb8(x) else x.shape[-1]
    assert input_size / (2 ** (self.max_depth - 1)) >= 1, \
        'Input size is {}. It must be >= {}'.format(input_size, 2 ** (self.max_depth - 1))
    assert num_powers_of_two(input_size) >= self.max_depth, \
        'Input size is {}. It must be a multiple of 2^(max_depth) = 2^{} = {}'.format(
            input_size, self.max_depth, 2 ** self.max_depth)
    dt = getattr(self, '_ms_dt', 0)
    was_plain = bool(dt) and not ops16.is_cb8(x)
    if was_plain:
      x = ops16.to_cb8(x, dt)
    channels = self.conv2[-1].conv.weight.shape[0]
    for i, m in enumerate(self.pre_downsampling_conv):
      x = m(x, _ms_chain=_chain_ok(self.pre_downsampling_conv, i))
    residuals = [x]
    # residuals[j] feeds conv1[j] AND, as the residual, conv2[max_depth - 1 - j]: the two gradients meet inside conv1[j]'s
    # data-gradient launch (ops.ResidualLink) instead of in an accumulation launch per level -- unless a hook could be looking
    links = []
    for i, down in enumerate(self.conv1):
      producer = self.conv1[i - 1] if i else self.pre_downsampling_conv[-1]
      up = self.conv2[self.max_depth - 1 - i]
      lk = ops.ResidualLink() if (not dt and torch.is_grad_enabled() and not _hooked(producer, down, up)) else None
      links.append(lk)
      x = down(x, _ms_link=(lk, 'consumer')) if lk is not None else down(x)
      if i < self.max_depth - 1:
        residuals.append(x)
    bn = x
    for i, up in enumerate(self.conv2):
      lk = links[self.max_depth - i - 1]
      x = up.forward_upsample_add(x, residuals[self.max_depth - i - 1], _ms_link=(lk, 'residual') if lk is not None else None)
    if was_plain:
      x, bn = ops16.from_cb8(x, channels), ops16.from_cb8(bn, channels)
    return (x, bn) if return_bottleneck else x


class AudioEncoder(nn.Module):
  '''
  input_shape:  (N, C, time, frequency)
  output_shape: (N, 256, output_feats)
  '''

  def __init__(self, output_feats=64, input_channels=1, kernel_size=None, stride=None, p=0, groups=1):
    super(AudioEncoder, self).__init__()
    self.conv = nn.ModuleList([])
    common = dict(type='2d', leaky=True, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    for cin, cout, down in ((input_channels, 64, False), (64, 64, True), (64, 128, False), (128, 128, True),
                            (128, 256, False), (256, 256, True), (256, 256, False)):
      self.conv.append(ConvNormRelu(cin, cout, downsample=down, **common))
    self.conv.append(ConvNormRelu(256, 256, type='2d', leaky=True, downsample=False, kernel_size=(3, 8), stride=1,
                                  p=p, groups=groups))

  def forward(self, x, time_steps=None):
    if time_steps is None:
      time_steps = x.shape[-2]
    dt = getattr(self, '_ms_dt', 0)
    if dt:
      x = ops16.to_cb8(x, dt)
    for m in self.conv:
      x = m(x)
    if dt:
      x = ops16.from_cb8(x, self.conv[-1].conv.weight.shape[0])      # the resize runs on the fp32 kernel
    return ops.lerp_time(x, time_steps)


class _TimeMajorStack(nn.Module):
  """(N, time, feats) -> channel-major -> a chain of 1-D ConvNormRelu blocks."""

  def _build(self, specs, kernel_size, stride, p, groups):
    self.conv = nn.ModuleList([])
    for cin, cout, down in specs:
      self.conv.append(ConvNormRelu(cin, cout, type='1d', leaky=True, downsample=down, kernel_size=kernel_size,
                                    stride=stride, p=p, groups=groups))

  def _chain(self, x, last_f32=False):
    """16-bit modes: the result is a plain fp32 (B, C, T') tensor either way (converted, or written so by the last block)."""
    dt = getattr(self, '_ms_dt', 0)
    if dt:
      x = ops16.btc_to_cb8(x, dt)
      mods = list(self.conv)
      for m in mods[:-1]:
        x = m(x)
      if last_f32:
        return mods[-1](x, _out_f32=True)
      return ops16.from_cb8(mods[-1](x), mods[-1].conv.weight.shape[0])
    x = ops.to_channel_major(x)
    for i, m in enumerate(self.conv):
      x = m(x, _ms_chain=_chain_ok(self.conv, i))
    return x


def _same_res_specs(cin):
  return ((cin, 64, False), (64, 64, False), (64, 128, False), (128, 128, False), (128, 256, False),
          (256, 256, False))


class PoseEncoder(_TimeMajorStack):
  '''
  input_shape:  (N, time, pose_features)
  output_shape: (N, 256, time)
  '''

  def __init__(self, output_feats=64, input_channels=96, kernel_size=None, stride=None, p=0, groups=1):
    super(PoseEncoder, self).__init__()
    self._build(_same_res_specs(input_channels), kernel_size, stride, p, groups)

  def forward(self, x, time_steps=None):
    return self._chain(x).squeeze(dim=-1)


class TextEncoder1D(_TimeMajorStack):
  '''
  input_shape:  (N, time, text_features: 300)
  output_shape: (N, 256, time)
  '''

  def __init__(self, output_feats=64, input_channels=300, kernel_size=None, stride=None, p=0, groups=1):
    super().__init__()
    self._build(_same_res_specs(input_channels), kernel_size, stride, p, groups)

  def forward(self, x, time_steps=None, **kwargs):
    return self._chain(x).squeeze(dim=-1)


class PoseStyleEncoder(_TimeMajorStack):
  '''
  input_shape:  (N, time, pose_features)
  output_shape: (N, num_speakers)
  '''

  def __init__(self, output_feats=64, input_channels=96, kernel_size=None, stride=None, p=0, groups=1,
               num_speakers=4):
    super().__init__()
    self._build(((input_channels, 64, False), (64, 64, True), (64, 128, True), (128, 128, True), (128, 256, True),
                 (256, 256, True), (256, num_speakers, True)), kernel_size, stride, p, groups)

  def forward(self, x, time_steps=None):
    x = self._chain(x, last_f32=True)
    x = x.mean(-1) if x.shape[-1] > 1 else x.squeeze(-1)   # mean over a length-1 axis is the identity
    return x.squeeze(dim=-1)


class ClusterClassify(nn.Module):
  '''
  input_shape: (B, C, T)
  output_shape: (B, num_clusters, T)
  '''

  def __init__(self, num_clusters=8, kernel_size=None, stride=None, p=0, groups=1, input_channels=256):
    super().__init__()
    self.conv = nn.ModuleList()
    common = dict(type='1d', leaky=True, downsample=False, kernel_size=kernel_size, stride=stride, p=p, groups=groups)
    self.conv.append(ConvNormRelu(input_channels, 256, **common))
    self.conv += nn.ModuleList([ConvNormRelu(256, 256, **common) for i in range(5)])
    self.logits = nn.Conv1d(256 * groups, num_clusters * groups, kernel_size=1, stride=1, groups=groups)

  def forward(self, x, time_steps=None):
    dt = getattr(self, '_ms_dt', 0)
    if dt and not ops16.is_cb8(x):
      x = ops16.to_cb8(x, dt)
    for i, m in enumerate(self.conv):
      x = m(x, _ms_chain=_chain_ok(self.conv, i))
    return bare_conv(self.logits, x, out_f32=True)        # scores (B, M, T) are fp32 in every mode


class Group(nn.Module):
  """layers.py:593-650.  Constructed by the generator (state_dict keys style_dec_gr.*) but never called on the audio path."""

  def __init__(self, models, groups=1, dim=1):
    super().__init__()
    if not isinstance(models, list):
      models = [models]
    self.models = nn.ModuleList(models)
    self.groups = groups
    self.dim = dim

  @staticmethod
  def mix_groups(z, weights, groups):
    """Soft mixture over the `groups` channel slices of z (B, groups*F, T) with per-frame weights (B, T, groups) ->
    (B, T, F): out[b,t,f] = sum_g weights[b,t,g] * z[b, g*F+f, t]  (layers.py:618-627)."""
    B, C, T = z.shape
    w = weights.reshape(B, T, groups)
    return torch.einsum('bgft,btg->btf', z.reshape(B, groups, C // groups, T), w)

  def index_select_outputs(self, x, labels):
    return self.mix_groups(x, labels, self.groups)

  def forward(self, x, labels=None, transpose=True, **kwargs):
    """layers.py:629-650.  Off the audio path (plain torch ops; the sub-modules run whatever kernels they own): the inputs
    are joined along `dim`, optionally moved to channel-major, passed through the models; with `labels` the groups are mixed
    per frame, otherwise the result is handed back as one tensor per group."""
    if self.dim == 0:
      self.groups = len(x)
    if isinstance(x, (list, tuple)):
      x = torch.cat(list(x), dim=self.dim)
    if transpose:
      x = x.transpose(-1, -2)
    for model in self.models:
      x = model(x, **kwargs) if kwargs else model(x)
    if labels is not None:
      return self.mix_groups(x, labels, self.groups).transpose(-1, -2)
    return list(torch.chunk(x, self.groups, dim=self.dim % x.dim()))


class EmbLin(nn.Module):
  def __init__(self, num_embeddings, embedding_dim):
    super().__init__()
    self.num_embeddings = num_embeddings
    self.embedding_dim = embedding_dim
    self.emb = nn.Embedding(num_embeddings, embedding_dim)

  def forward(self, x, mode='lin'):
    if mode == 'lin':
      return x.matmul(self.emb.weight)
    elif mode == 'emb':
      return torch.nn.functional.embedding(x, self.emb.weight)


class Curriculum():
  def __init__(self, start, end, num_iters):
    self.start = start
    self.end = end
    self.num_iters = num_iters
    self.iters = 0
    self.diff = (end - start) / num_iters
    self.value = start

  def step(self, flag=True):
    if not flag:
      return self.value
    if self.iters >= self.num_iters:
      return self.end
    previous = self.value
    self.value += self.diff
    self.iters += 1
    return previous
