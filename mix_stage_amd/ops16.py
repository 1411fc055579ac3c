"""torch.autograd bindings of the 16-bit arithmetic mode (bf16 / fp16 operands, fp32 accumulate; include/mixstage.h: ms_dtype).

Activations travel between the blocks as "cb8" tensors: torch tensors of dtype bfloat16 / float16 and shape
(B, C8, T, 8) or (B, C8, H, W, 8), C8 = ceil(C/8) -- [channel block][pixel][8 channels], pad channels zero -- the layout the
matrix cores consume without transposition (mix_stage_amd/csrc/conv16.h).  Parameters, BatchNorm statistics, losses and
parameter gradients stay fp32.  The converters below sit at the fp32 boundaries of the path (inputs, scores, poses).
"""
import ctypes

import torch

from . import ops
from ._lib import (BwdOptions, ConvDesc, FwdOptions, MS_BARE, MS_BF16, MS_BN_EVAL, MS_BN_TRAIN, MS_DT_BN_FOLDED, MS_DT_OUT_F32, MS_DT_STAT_PAIR,
                   MS_F16, MS_IN_BCAST, MS_IN_PLAIN, MS_IN_UP2ADD, MS_LRELU, Prep16Item, check, lib)
from .ops import _grad_slot, _ptr, _stream, workspace

import os as _os
TORCH_DT = {MS_BF16: torch.bfloat16, MS_F16: torch.float16}
MS_DT = {torch.bfloat16: MS_BF16, torch.float16: MS_F16}
NAME_DT = {'bf16': MS_BF16, 'bfloat16': MS_BF16, 'fp16': MS_F16, 'f16': MS_F16, 'float16': MS_F16, 'half': MS_F16}


def is_cb8(t):
  return t is not None and t.dtype in MS_DT and t.dim() in (4, 5) and t.shape[-1] == 8


def _need16(*tensors):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise ops._lib.MixStageLibError('mix_stage_amd ops run on the MI355X only (got a %s tensor)' % t.device)


# ------------------------------------------------------------------------------------------------
# In-launch BatchNorm / chained decoder (ms_fwd_options.bn_sync): the arrival counters through which the workgroups of a launch
# exchange their batch statistics.  One persistent zeroed buffer per (device, stream): launches on one stream are serialised and
# leave the counters re-armed; two streams never share counters.  Word 0 is raised when a workgroup gave up waiting for its
# peers -- the block's output is then NaN and bn_sync_error() reports it (MixStageTrainStep.check_health raises).
# The in-launch forms assume the launch has the device to itself (all its workgroups resident at once): a second process on the
# same GPU, or a large kernel on another stream, can make a launch wait until its bound expires.
_bn_sync = {}
BN_SYNC_WORDS = 1 << 16


_in_launch = {'on': True}


def set_in_launch_meetings(on):
  """False: no launch of this process waits for its own workgroups (in-launch BatchNorm, chained decoder) -- for set-ups in which a
  launch does not have the device to itself (several ranks on one GPU)."""
  _in_launch['on'] = bool(on)


def in_launch_meetings():
  return _in_launch['on']


def _ensure_bn_sync(device):
  if not _in_launch['on']:
    return None
  key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
  buf = _bn_sync.get(key)
  if buf is None:
    buf = _bn_sync[key] = torch.zeros(BN_SYNC_WORDS, dtype=torch.int32, device=device)
  return buf


def chain_sync(device, B, M, words):
  """The chained decoder's meeting counters: a zeroed buffer per (device, stream, B, M) -- the counters are monotonic, which needs
  every launch that shares them to have the same member counts."""
  if not _in_launch['on']:
    return None
  key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, 'chain', B, M)
  buf = _bn_sync.get(key)
  if buf is None or buf.numel() < words:
    buf = _bn_sync[key] = torch.zeros(words, dtype=torch.int32, device=device)
  return buf


def block_sync(device, desc, words=1024, tag=''):
  """Meeting counters of ONE conv block (fp32 clip-resident kernels): a zeroed buffer per (device, stream, block descriptor) -- the
  counters are monotonic, so every launch that shares them must have the same member counts: one block, one shape."""
  if not _in_launch['on']:
    return None
  key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream, 'block', id(desc), tag)
  buf = _bn_sync.get(key)
  if buf is None:
    buf = _bn_sync[key] = torch.zeros(words, dtype=torch.int32, device=device)
    _block_sync_keep.append(desc)          # (id(desc) stays unique while the buffer lives)
  return buf


_block_sync_keep = []


def bn_sync_error():
  """True when an in-launch BatchNorm workgroup timed out waiting for its group (synchronises the device)."""
  return any(int(b[0].item()) != 0 for b in _bn_sync.values())


def bn_sync_words():
  """Diagnostics: the first words of every sync buffer (word 0 = error code)."""
  return [b[:3].tolist() for b in _bn_sync.values()]


def bn_sync_clear():
  for b in _bn_sync.values():
    b.zero_()


def check_meetings():
  """Raises if any launch of this process whose workgroups meet inside the launch gave up waiting (its outputs were NaN), and
  re-arms the counters.  Synchronises the device."""
  if bn_sync_error():
    words = bn_sync_words()
    bn_sync_clear()
    raise RuntimeError('(sync words %s) an in-launch BatchNorm meeting timed out (the launch did not have the GPU to itself): the launch produced '
                       'NaN; see MixStageTrainStep.check_health' % (words,))


# ------------------------------------------------------------------------------------------------
# layout converters
class _ToCb8Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, dt):
    _need16(x)
    if x.dtype != torch.float32:
      raise TypeError('to_cb8 expects a float32 (B, C, ...) tensor, got %s' % x.dtype)
    x = x.contiguous()
    B, C = x.shape[0], x.shape[1]
    sp = tuple(x.shape[2:])
    hw = 1
    for v in sp:
      hw *= v
    y = torch.empty((B, (C + 7) // 8) + sp + (8,), dtype=TORCH_DT[dt], device=x.device)
    check(lib().ms_cb8_from_plain(dt, _ptr(x), _ptr(y), B, C, hw, _stream()), 'ms_cb8_from_plain')
    ctx.meta = (dt, B, C, sp, hw)
    return y

  @staticmethod
  def backward(ctx, dy):
    dt, B, C, sp, hw = ctx.meta
    dy = dy.contiguous()
    dx = torch.empty((B, C) + sp, dtype=torch.float32, device=dy.device)
    check(lib().ms_cb8_to_plain(dt, _ptr(dy), _ptr(dx), B, C, hw, _stream()), 'ms_cb8_to_plain')
    return dx, None


class _FromCb8Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, C):
    _need16(x)
    x = x.contiguous()
    dt = MS_DT[x.dtype]
    B = x.shape[0]
    sp = tuple(x.shape[2:-1])
    hw = 1
    for v in sp:
      hw *= v
    y = torch.empty((B, C) + sp, dtype=torch.float32, device=x.device)
    check(lib().ms_cb8_to_plain(dt, _ptr(x), _ptr(y), B, C, hw, _stream()), 'ms_cb8_to_plain')
    ctx.meta = (dt, B, C, sp, hw, tuple(x.shape))
    return y

  @staticmethod
  def backward(ctx, dy):
    dt, B, C, sp, hw, shape = ctx.meta
    dy = dy.contiguous()
    dx = torch.empty(shape, dtype=TORCH_DT[dt], device=dy.device)
    check(lib().ms_cb8_from_plain(dt, _ptr(dy), _ptr(dx), B, C, hw, _stream()), 'ms_cb8_from_plain')
    return dx, None


class _BtcToCb8Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, dt, velocity):
    _need16(x)
    if x.dtype != torch.float32:
      raise TypeError('btc_to_cb8 expects a float32 (B, T, C) tensor, got %s' % x.dtype)
    x = x.contiguous()
    B, T, C = x.shape
    y = torch.empty((B, (C + 7) // 8, T, 8), dtype=TORCH_DT[dt], device=x.device)
    check(lib().ms_cb8_from_btc(dt, _ptr(x), _ptr(y), B, T, C, 1 if velocity else 0, _stream()), 'ms_cb8_from_btc')
    ctx.meta = (dt, B, T, C, velocity)
    return y

  @staticmethod
  def backward(ctx, dy):
    dt, B, T, C, velocity = ctx.meta
    dy = dy.contiguous()
    dx = torch.empty((B, T, C), dtype=torch.float32, device=dy.device)
    check(lib().ms_cb8_to_btc(dt, _ptr(dy), _ptr(dx), B, T, C, 1 if velocity else 0, _stream()), 'ms_cb8_to_btc')
    return dx, None, None


def to_cb8(x, dt):
  """fp32 (B, C, ...) channel-major -> cb8 (B, C8, ..., 8)."""
  return _ToCb8Fn.apply(x, int(dt))


def from_cb8(x, C):
  """cb8 (B, C8, ..., 8) -> fp32 (B, C, ...) channel-major."""
  return _FromCb8Fn.apply(x, int(C))


def btc_to_cb8(x, dt, velocity=False):
  """fp32 time-major (B, T, C) -> cb8 (B, C8, T, 8); velocity=True fuses GAN.get_velocity (gan.py:47-52)."""
  return _BtcToCb8Fn.apply(x, int(dt), bool(velocity))


# ------------------------------------------------------------------------------------------------
# prepared operands (ms_weights16_prepare): once per optimizer update for all blocks of a network in one launch
def _prepare16(entries):
  entries = [e for e in entries if e['n']]
  if not entries:
    return
  n = len(entries)
  items = (Prep16Item * n)()
  for it, e in zip(items, entries):
    it.desc = ctypes.pointer(e['d'])
    it.w = e['w'].data_ptr()
    for k in ('bias', 'gamma', 'beta', 'running_mean', 'running_var'):
      t = e.get(k)
      setattr(it, k, t.data_ptr() if t is not None else None)
    it.fwd = e['wt'].data_ptr() if e['kind'] == 'fwd16' else None
    it.dgrad = e['wt'].data_ptr() if e['kind'] == 'dgrad16' else None
  check(lib().ms_weights16_prepare(n, items, _stream()), 'ms_weights16_prepare')
  tune = lib().ms_tuning_epoch()
  for e in entries:
    e['version'], e['tune'] = e['w']._version, tune


def _prepared16_for(w, d, kind, bn=None):
  """The block's prepared 16-bit operand (None: feature off): kind 'fwd16' / 'dgrad16'."""
  P = ops._prepared
  if not P['on']:
    return None
  if kind == 'fwd16' and d.mode == MS_BN_EVAL and not (d.dtype & MS_DT_BN_FOLDED):
    pass      # unfolded eval blocks share the training weights: the epilogue applies the running statistics
  key = (w.data_ptr(), id(d), kind)
  e = P['entries'].get(key)
  if e is None:
    n = lib().ms_weights16_bytes(ctypes.byref(d), 0 if kind == 'fwd16' else 1)
    e = dict(w=w, d=d, n=(n + 3) // 4, wt=None, version=-1, tune=-1, kind=kind)
    if bn is not None:
      e.update(bn)
    if e['n']:
      e['wt'] = torch.empty(e['n'], dtype=torch.float32, device=w.device)
    P['entries'][key] = e
    P['by_storage'].setdefault(w.untyped_storage().data_ptr(), []).append(e)
  if not e['n']:
    return None
  if e['version'] != w._version:
    _prepare16([e])
  return e['wt']


# ------------------------------------------------------------------------------------------------
class _ConvBlock16Fn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, x2, w, bias, gamma, beta, geom, mode, in_mode, stats, dt_flags, pre=None):
    rm, rv = stats if stats is not None else (None, None)
    _need16(x, x2, w, bias, gamma, beta, rm, rv)
    if not is_cb8(x) or (x2 is not None and not is_cb8(x2)):
      raise TypeError('16-bit conv block expects cb8 inputs (ops16.to_cb8), got %s %s' % (x.dtype, tuple(x.shape)))
    dt = MS_DT[x.dtype]
    if (dt_flags & 0xff) != dt:
      raise TypeError('block is set to dtype %d but the input is %s' % (dt_flags & 0xff, x.dtype))
    x = x.contiguous()
    x2 = x2.contiguous() if x2 is not None else None
    nd = geom.nd
    if x.dim() != nd + 3:
      raise RuntimeError('conv block expects a %d-D cb8 input, got %s' % (nd + 3, tuple(x.shape)))
    B = x.shape[0]
    ref = x2 if in_mode == MS_IN_UP2ADD else x
    H, W = (ref.shape[2], ref.shape[3]) if nd == 2 else (1, ref.shape[2])
    ctot = w.shape[0]
    Cout_g = ctot // geom.groups
    Cin_g = w.shape[1]
    exp_c = Cin_g if in_mode == MS_IN_BCAST else Cin_g * geom.groups
    if x.shape[1] != (exp_c + 7) // 8:
      raise RuntimeError('conv block: expected %d input channel blocks, got %d' % ((exp_c + 7) // 8, x.shape[1]))
    if in_mode == MS_IN_UP2ADD and (x.shape[2] * 2 != W or x2.shape[1] != x.shape[1]):
      raise RuntimeError('UP2ADD: a %s and residual %s do not match' % (tuple(x.shape), tuple(x2.shape)))
    d = geom.desc(B, Cin_g, H, W, Cout_g, mode, in_mode, dt_flags)
    out_f32 = bool(dt_flags & MS_DT_OUT_F32)
    sp = (d.OH, d.OW) if nd == 2 else (d.OW,)
    c8 = (ctot + 7) // 8
    if pre is not None:
      # results produced by the chained decoder launch (decoder_chain16): this node only carries the block's backward pass
      y_raw, y, save = pre
    else:
      if out_f32:
        y = torch.empty((B, ctot) + sp, dtype=torch.float32, device=x.device)
      else:
        y = torch.empty((B, c8) + sp + (8,), dtype=x.dtype, device=x.device)
      y_raw = save = None
      if mode == MS_BN_TRAIN:
        y_raw = torch.empty((B, c8) + sp + (8,), dtype=x.dtype, device=x.device)
        save = torch.empty((8 if (dt_flags & MS_DT_STAT_PAIR) else 4) * ctot, dtype=torch.float32, device=x.device)
      ws = workspace(d._fwd_ws, x.device)
      sync = _ensure_bn_sync(x.device) if mode == MS_BN_TRAIN else None
      folded = mode == MS_BN_EVAL and bool(dt_flags & MS_DT_BN_FOLDED)
      planes = _prepared16_for(w, d, 'fwd16', dict(bias=bias, gamma=gamma, beta=beta, running_mean=rm, running_var=rv)
                               if folded else None)
      opt = FwdOptions(planes.data_ptr() if planes is not None else None, sync.data_ptr() if sync is not None else None,
                       sync.numel() if sync is not None else 0)
      check(lib().ms_conv_block_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta),
                                       _ptr(rm), _ptr(rv), _ptr(y_raw), _ptr(y), _ptr(save), _ptr(ws), ws.numel(),
                                       _stream(), ctypes.byref(opt)), 'ms_conv_block_fwd_ex')
    ctx.geom_desc = d
    ctx.mode, ctx.in_mode = mode, in_mode
    ctx.has_bias = bias is not None
    ctx.params = (w, bias, gamma, beta)
    ctx.raw_shape = (B, c8) + sp + (8,)
    # (BN_TRAIN blocks with a cb8 output: the backward pass reads x_hat and the activation mask from y wherever BatchNorm +
    # LeakyReLU invert safely -- y_raw is only written for the channel blocks where they do not, csrc/conv16.h)
    ctx.save_for_backward(x, x2, w, gamma, y_raw, y if (mode == MS_LRELU or (mode == MS_BN_TRAIN and not out_f32)) else None, save)
    return y

  @staticmethod
  def backward(ctx, dy):
    x, x2, w, gamma, y_raw, y, save = ctx.saved_tensors
    d, mode, in_mode = ctx.geom_desc, ctx.mode, ctx.in_mode
    if mode == MS_BN_EVAL:
      raise RuntimeError('backward through an eval-mode (running-stats) ConvNormRelu is not on the path')
    need_x, need_x2, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    need_bn = mode == MS_BN_TRAIN and ctx.needs_input_grad[4]
    dy = dy.contiguous()
    dev = dy.device
    out_f32 = bool(d.dtype & MS_DT_OUT_F32)
    dyr = torch.empty(ctx.raw_shape, dtype=x.dtype, device=dev) if (mode != MS_BARE or out_f32) else None
    up2 = in_mode == MS_IN_UP2ADD
    want_dx = need_x or (up2 and need_x2)
    dx = torch.empty_like(x) if want_dx else None
    dx2 = torch.empty_like(x2) if (want_dx and up2) else None
    pw, pbias, pgamma, pbeta = ctx.params
    dw = dbias = dgamma = dbeta = None
    direct_w = direct_b = direct_g = direct_be = False
    if need_w:
      dw, direct_w = _grad_slot(pw, w)
      if ctx.has_bias:
        dbias, direct_b = _grad_slot(pbias, pbias)
    if need_bn:
      dgamma, direct_g = _grad_slot(pgamma, gamma)
      dbeta, direct_be = _grad_slot(pbeta, gamma)
    ws = workspace(d._bwd_ws, dev)
    wt = _prepared16_for(w, d, 'dgrad16') if want_dx else None
    part, nsplit = (None, 1)
    D = ops._deferred
    if D['on'] and direct_w is True:
      part, nsplit = ops._wgrad_partials_for(w, d)
    # the weight-gradient kernel itself is queued too when nothing downstream reads dw (it lands in the flat buffer)
    defer_launch = bool(D['on'] and direct_w is True and dw is not None and ops.DEFER_WGRAD_LAUNCH)
    opt = BwdOptions(None, None, 0, wt.data_ptr() if wt is not None else None, part.data_ptr() if part is not None else None,
                     1 if defer_launch else 0)
    check(lib().ms_conv_block_bwd_ex(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(gamma), None, None, _ptr(y_raw),
                                     _ptr(y), _ptr(save), _ptr(dy), _ptr(dyr), _ptr(dx), _ptr(dx2), _ptr(dw),
                                     _ptr(dbias), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), _stream(),
                                     ctypes.byref(opt)), 'ms_conv_block_bwd_ex')
    if defer_launch:
      D['launches'] += 1
      D['keep'].append((x, x2, dy, dyr, dw, part))
      ops._queue_deferred_flush()
    if part is not None:
      D['jobs'].append((part, dw, nsplit))
      ops._queue_deferred_flush()
    return (dx, dx2, None if direct_w else dw, None if direct_b else dbias, None if direct_g else dgamma,
            None if direct_be else dbeta, None, None, None, None, None, None)


def conv_block16(x, w, bias, geom, mode, gamma=None, beta=None, running_mean=None, running_var=None, x2=None,
                 in_mode=MS_IN_PLAIN, out_f32=False, bn_folded=False):
  """One conv block in the 16-bit mode: x (and x2) cb8, result cb8 -- or plain fp32 (B, C, ...) with out_f32."""
  stats = (running_mean, running_var) if running_mean is not None else None
  flags = MS_DT[x.dtype] | (MS_DT_OUT_F32 if out_f32 else 0) | (MS_DT_BN_FOLDED if (bn_folded and mode == MS_BN_EVAL) else 0)
  if ops.stat_pair_active():
    flags |= MS_DT_STAT_PAIR            # two passes of the module side by side in this batch (ops.stat_pair)
  return _ConvBlock16Fn.apply(x, x2, w, bias, gamma, beta, geom, mode, in_mode, stats, flags)


# ------------------------------------------------------------------------------------------------
def decoder_chain16(x, blocks, logits, score, P):
  """16-bit form of ops.decoder_chain: x cb8 (B, 34, T, 8) shared by all groups; (out (B,T,P) fp32, soft (B,T,M)) or None."""
  if not ops.USE_DECODER_CHAIN or len(blocks) != 4 or not is_cb8(x) or x.dim() != 4 or ops.bn_sync_active() or not in_launch_meetings():
    return None
  dt = MS_DT[x.dtype]
  blk0 = blocks[0]
  M = blk0.conv.groups
  B, cb0, T = x.shape[0], x.shape[1], x.shape[2]
  cin0 = blk0.conv.weight.shape[1]
  if (cin0 + 7) // 8 != cb0:
    return None
  if any(getattr(m, '_ms_dt', 0) != dt for m in blocks) or getattr(logits, '_ms_dt', 0) != dt or any(m._p and m.training for m in blocks):
    return None
  if any(getattr(m, '_bn_folded', False) for m in blocks):
    return None
  if any(m._forward_hooks or m._forward_pre_hooks for m in list(blocks) + [logits]):
    return None
  if any(m.conv.groups != M or m.conv.kernel_size != (3,) or m.conv.stride != (1,) or m.conv.padding != (1,) or
         m.conv.weight.shape[0] != 256 * M or m.conv.weight.dtype != torch.float32 or m._slope != blk0._slope for m in blocks):
    return None
  if any(m.conv.weight.shape[1] != 256 for m in blocks[1:]):
    return None
  if logits.groups != M or logits.kernel_size != (1,) or logits.weight.shape[0] != M * P or logits.weight.shape[1] != 256 or logits.bias is None:
    return None
  if tuple(score.shape) != (B, M, T) or score.dtype != torch.float32:
    return None
  training = all(m.training and m.norm.track_running_stats for m in blocks)
  if not training and any(m.training for m in blocks):
    return None
  mode = MS_BN_TRAIN if training else MS_BN_EVAL
  params = [t for m in blocks for t in (m.conv.weight, m.conv.bias, m.norm.weight, m.norm.bias)] + [logits.weight, logits.bias]
  need_grad = torch.is_grad_enabled() and (x.requires_grad or score.requires_grad or any(t is not None and t.requires_grad for t in params))
  if need_grad and not training:
    return None
  d = ops._chain_desc(B, M, T, cin0, P, mode, blk0, dt)
  if not lib().ms_decoder_chain_supported(ctypes.byref(d)):
    return None
  _need16(x, score, *[t for t in params if t is not None])
  x, score = x.contiguous(), score.contiguous()
  dev = x.device
  wts = [m.conv.weight for m in blocks] + [logits.weight]
  prepared = ops._chain_prepared(d, wts)
  sync = chain_sync(dev, B, M, ops.CHAIN_SYNC_FIRST_WORD + lib().ms_decoder_chain_sync_words(ctypes.byref(d)))
  C = 256 * M
  keep = need_grad
  act = lambda: torch.empty((B, C // 8, T, 8), dtype=x.dtype, device=dev)
  y_raw = [act() if keep else None for _ in blocks]
  y = [act() if keep else None for _ in blocks]
  save = [torch.empty(4 * C, dtype=torch.float32, device=dev) if keep else None for _ in blocks]
  z = torch.empty((B, M * P, T), dtype=torch.float32, device=dev) if keep else None
  soft = torch.empty((B, T, M), dtype=torch.float32, device=dev)
  out = torch.empty((B, T, P), dtype=torch.float32, device=dev)
  tn = ops._lib.ChainTensors()
  tn.x, tn.score = x.data_ptr(), score.data_ptr()
  for l, m in enumerate(blocks):
    tn.w[l], tn.bias[l] = m.conv.weight.data_ptr(), (m.conv.bias.data_ptr() if m.conv.bias is not None else None)
    tn.gamma[l], tn.beta[l] = m.norm.weight.data_ptr(), m.norm.bias.data_ptr()
    tn.running_mean[l], tn.running_var[l] = m.norm.running_mean.data_ptr(), m.norm.running_var.data_ptr()
    tn.y_raw[l] = y_raw[l].data_ptr() if keep else None
    tn.y[l] = y[l].data_ptr() if keep else None
    tn.save[l] = save[l].data_ptr() if keep else None
  tn.w_logits, tn.bias_logits = logits.weight.data_ptr(), logits.bias.data_ptr()
  tn.z = z.data_ptr() if keep else None
  tn.soft, tn.out, tn.prepared = soft.data_ptr(), out.data_ptr(), prepared.data_ptr()
  tn.sync, tn.sync_words = sync.data_ptr(), sync.numel()
  wsp = workspace(lib().ms_decoder_chain_workspace(ctypes.byref(d)), dev)
  check(lib().ms_decoder_chain_fwd(ctypes.byref(d), ctypes.byref(tn), _ptr(wsp), wsp.numel(), _stream()), 'ms_decoder_chain_fwd')
  if training:
    for m in blocks:
      m._note_train_pass()
  if not need_grad:
    return out, soft
  h = x
  for l, m in enumerate(blocks):
    n = m.norm
    h = _ConvBlock16Fn.apply(h, None, m.conv.weight, m.conv.bias, n.weight, n.bias, m._geometry(), MS_BN_TRAIN,
                             MS_IN_BCAST if l == 0 else MS_IN_PLAIN, (n.running_mean, n.running_var), dt, (y_raw[l], y[l], save[l]))
  geom = getattr(logits, '_ms_geom', None)
  if geom is None:
    geom = logits._ms_geom = ops.ConvGeom(1, logits.groups, logits.kernel_size, logits.stride, logits.padding, slope=0.0)
  zt = _ConvBlock16Fn.apply(h, None, logits.weight, logits.bias, None, None, geom, MS_BARE, MS_IN_PLAIN, None, dt | MS_DT_OUT_F32,
                            (None, z, None))
  return ops._SoftmaxMixFn.apply(zt, score, int(P), (out, soft))
