"""torch.autograd bindings of the HIP kernels (libmixstage_hip.so, include/mixstage.h).

PyTorch is plumbing here: it owns HBM allocations, the current HIP stream and the autograd tape.
Every function below enqueues hand-written gfx950 kernels through the C-ABI with raw device
pointers; there is no eager/CPU fallback -- a CPU tensor or a missing library raises.
"""
import contextlib
import ctypes
import os

import torch

from . import _lib
from ._lib import (BwdOptions, ConvDesc, FwdOptions, MS_BARE, MS_BN_EVAL, MS_BN_TRAIN, MS_DT_STAT_PAIR, MS_IN_BCAST, MS_IN_PLAIN,
                   MS_IN_UP2ADD, MS_LRELU, check, lib)

_vp = ctypes.c_void_p


def _ptr(t):
  return None if t is None else _vp(t.data_ptr())


def _stream():
  return _vp(torch.cuda.current_stream().cuda_stream)


def _need_hip(*tensors):
  for t in tensors:
    if t is None:
      continue
    if not t.is_cuda:
      raise _lib.MixStageLibError('mix_stage_amd ops run on the MI355X only (got a %s tensor); there is no CPU '
                                  'fallback' % t.device)
    if t.is_floating_point() and t.dtype != torch.float32:
      raise TypeError('the HIP path computes in float32 (got %s); call .float() on the model and inputs' % t.dtype)


_workspaces = {}
_counters = {}
# In-launch "last arriver" reduction of split-K / split-pixel slabs (ms_set_counter_buffer).  Correct and bitwise
# reproducible, but measured SLOWER on MI355X (5609 vs 6878 clips/s): every workgroup pays an agent-scope release fence
# (L2 write-back, 2-6 us) at its tail, more than the ~5 us reduce kernel it saves.  Off.
USE_IN_LAUNCH_SPLIT_REDUCTION = False


def _ensure_counters(device):
  """Persistent zeroed arrival counters for the in-launch split reductions (ms_set_counter_buffer)."""
  key = (device.type, device.index)
  if key not in _counters:
    buf = torch.zeros(1 << 16, dtype=torch.int32, device=device)
    check(lib().ms_set_counter_buffer(_ptr(buf), buf.numel()), 'ms_set_counter_buffer')
    _counters[key] = buf
  return _counters[key]



def workspace(nbytes, device):
  """Scratch of the CURRENT stream on `device`, grown geometrically.  Users on one stream are serialised by it; two streams
  that run blocks concurrently get a buffer each (partial tiles, BatchNorm partials and split reductions live here)."""
  key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
  if USE_IN_LAUNCH_SPLIT_REDUCTION:
    _ensure_counters(device)
  ws = _workspaces.get(key)
  if ws is None or ws.numel() < nbytes:
    size = max(int(nbytes * 1.5), 1 << 22)
    if ws is not None:
      _retired_workspaces.append(ws)       # captured HIP graphs may still launch kernels that point into it
    ws = torch.empty(size, dtype=torch.uint8, device=device)
    _workspaces[key] = ws
  return ws


_retired_workspaces = []


_side_workspaces = {}


def side_workspace(nbytes, device):
  key = (device.type, device.index)
  ws = _side_workspaces.get(key)
  if ws is None or ws.numel() < nbytes:
    ws = torch.empty(max(int(nbytes * 1.5), 1 << 22), dtype=torch.uint8, device=device)
    _side_workspaces[key] = ws
  return ws


# Backward overlap: weight gradients run on a side HIP stream, concurrently with the data-gradient chain.
_overlap = {'stream': None, 'keep': []}


def set_backward_overlap(side_stream):
  """side_stream: a torch.cuda.Stream (or None to disable).  The caller must call join_backward_overlap() after the
  backward pass and before anything reads the weight gradients."""
  _overlap['stream'] = side_stream
  _overlap['keep'].clear()


def join_backward_overlap():
  side = _overlap['stream']
  if side is not None:
    torch.cuda.current_stream().wait_stream(side)
  _overlap['keep'].clear()


# ------------------------------------------------------------------------------------------------
# prepared data-gradient weights (ms_dgrad_weights_prepare): built once per optimizer update for all blocks of a network in
# ONE launch instead of one transposed copy per block and backward call.  Off unless a trainer that owns the parameter
# updates (MixStageTrainStep) switches it on and calls refresh_prepared_weights() after every update.
_prepared = {'on': False, 'entries': {}, 'by_storage': {}}


def enable_prepared_weights(on):
  _prepared['on'] = bool(on)
  if not on:
    _prepared['entries'].clear()
    _prepared['by_storage'].clear()


def _prepare_entries(entries):
  entries = [e for e in entries if e['n']]
  if not entries:
    return
  L = lib()
  tune = L.ms_tuning_epoch()
  e16 = [e for e in entries if e['kind'].endswith('16')]
  if e16:                                # 16-bit mode operands (ops16): one ms_weights16_prepare launch per 24 blocks
    from . import ops16
    ops16._prepare16(e16)
  for e in entries:
    if e['kind'] == 'chain32':           # the chained decoder's weight streams (decoder_chain): one launch for its five convs
      wl = (ctypes.c_void_p * 4)(*[x.data_ptr() for x in e['ws'][:4]])
      check(L.ms_decoder_chain_prepare(ctypes.byref(e['d']), wl, _ptr(e['ws'][4]), _ptr(e['wt']), _stream()), 'ms_decoder_chain_prepare')
      e['version'], e['tune'] = sum(x._version for x in e['ws']), tune
  for kind, fn in (('dgrad', L.ms_dgrad_weights_prepare), ('fwd', L.ms_fwd_weights_prepare)):
    es = [e for e in entries if e['kind'] == kind]
    if not es:
      continue
    n = len(es)
    descs = (ConvDesc * n)(*[e['d'] for e in es])
    ws = (ctypes.c_void_p * n)(*[e['w'].data_ptr() for e in es])
    outs = (ctypes.c_void_p * n)(*[e['wt'].data_ptr() for e in es])
    check(fn(n, descs, ws, outs, _stream()), 'ms_%s_weights_prepare' % kind)
    for e in es:
      e['version'], e['tune'] = e['w']._version, tune


def _prepared_for(w, d, kind='dgrad'):
  """The block's prepared buffer (None: not needed / feature off): kind 'dgrad' = the transposed (class-split) copy of w for
  the data gradient (+ its bf16 planes in bf16x6 mode), kind 'fwd' = the bf16 planes of w (bf16x6 mode only).  Builds or
  rebuilds it when w was modified in place by anything torch knows about (w._version) or a tuning knob moved."""
  if not _prepared['on']:
    return None
  L = lib()
  key = (w.data_ptr(), id(d), kind)
  tune = L.ms_tuning_epoch()
  e = _prepared['entries'].get(key)
  if e is None or e['tune'] != tune:
    if kind == 'dgrad':
      n = L.ms_dgrad_weights_elems(ctypes.byref(d), _ptr(w))           # floats
    else:
      n = (L.ms_fwd_weights_bytes(ctypes.byref(d)) + 3) // 4           # bytes -> floats
    if e is None:
      e = dict(w=w, d=d, n=0, wt=None, version=-1, tune=-1, kind=kind)
      _prepared['entries'][key] = e
      _prepared['by_storage'].setdefault(w.untyped_storage().data_ptr(), []).append(e)
    if n != e['n']:
      e['n'], e['wt'] = n, (torch.empty(n, dtype=torch.float32, device=w.device) if n else None)
    e['version'] = -1
    if not n:
      e['tune'] = tune
  if not e['n']:
    return None
  if e['version'] != w._version or e['tune'] != tune:
    _prepare_entries([e])
  return e['wt']


def drop_trainer_caches(storage_ptrs):
  """Forget the prepared weights and partial-gradient buffers of parameters living in these storages (a trainer that goes
  away calls this; the buffers are referenced by its captured graphs until then)."""
  for sp in storage_ptrs:
    for e in _prepared['by_storage'].pop(sp, ()):
      _prepared['entries'] = {k: v for k, v in _prepared['entries'].items() if v is not e}
  _deferred['bufs'] = {k: v for k, v in _deferred['bufs'].items() if v[2] not in storage_ptrs}


def refresh_prepared_weights(flat_params, start=0):
  """Rebuild every prepared buffer whose weight lives in `flat_params`' storage: one launch (per 48 blocks).  Call after
  each update of that storage that torch cannot see (the HIP Adam kernels write through raw pointers).  start: skip the
  first `start` buffers of that storage (those a captured graph already rebuilds, see prepared_count)."""
  if _prepared['on']:
    _prepare_entries(_prepared['by_storage'].get(flat_params.untyped_storage().data_ptr(), ())[start:])


def prepared_count(flat_params):
  """Number of prepared buffers registered for this storage so far (the list only grows): a graph captured now rebuilds
  exactly these; buffers that appear later (another step kind runs other blocks of the same network) need an eager
  refresh_prepared_weights(flat_params, start=count) after each replay."""
  return len(_prepared['by_storage'].get(flat_params.untyped_storage().data_ptr(), ()))


# ------------------------------------------------------------------------------------------------
# deferred weight-gradient reductions (ms_wgrad_reduce_multi): blocks that split dw's pixel reduction leave their partial
# slabs in per-block buffers; ONE launch at the end of the backward pass adds them all into the gradient slots.  Only for
# gradients written straight into the flat buffer (nobody downstream in autograd reads them).  Opt-in, like the above.
# In the 16-bit modes the weight-gradient KERNELS are deferred as well (ms_bwd_options.defer_wgrad_launch / ms_wgrad_flush):
# nothing in the backward chain reads them, and launched side by side in a few multi-block launches the small layers'
# one-wave kernels cost one launch latency instead of one each.  'keep' holds the tensors those kernels read until the flush.
_deferred = {'on': False, 'jobs': [], 'queued': False, 'bufs': {}, 'keep': [], 'launches': 0}
DEFER_WGRAD_LAUNCH = os.environ.get('MS_DEFER_WGRAD_LAUNCH', '1') != '0'   # experiments: MS_DEFER_WGRAD_LAUNCH=0 launches per block


def enable_deferred_wgrad(on):
  _deferred['on'] = bool(on)
  # planner hint (fp32 weight gradients): queued launches share the chip, so a layer's workgroups take long runs of the pixel
  # reduction instead of splitting it until the layer fills the chip alone (MS_WGRAD_TPS: tiles per workgroup, experiments)
  lib().ms_set_wgrad_batched(1 if (on and DEFER_WGRAD_LAUNCH) else 0, int(os.environ.get('MS_WGRAD_TPS', '0')))
  reset_deferred_wgrad()
  if not on:
    _deferred['bufs'].clear()


def reset_deferred_wgrad():
  """Drop jobs a failed backward pass may have left behind."""
  _deferred['jobs'].clear()
  _deferred['keep'].clear()
  _deferred['queued'] = False
  if _deferred['launches']:
    _deferred['launches'] = 0
    lib().ms_wgrad_discard()


def _queue_deferred_flush():
  """The queues (here and in the library: wgrad_patch.hip / wgrad16.hip) are process-wide and flushed on ONE stream at the end
  of ONE backward pass: every block that queues work during a backward pass must run on the device and stream of the first
  one.  (The queued kernels ACCUMULATE into the gradient slots: FlatAdam.zero_grad of the same step is a precondition.)"""
  owner = (torch.cuda.current_device(), _stream().value)
  if not _deferred['queued']:
    _deferred['queued'] = True
    _deferred['owner'] = owner
    torch.autograd.Variable._execution_engine.queue_callback(_flush_deferred_wgrad)
  elif _deferred.get('owner') != owner:
    raise _lib.MixStageLibError('deferred weight-gradient work was queued from two devices or streams in one backward pass '
                                '(%s, then %s): run concurrent trainers in separate processes or with '
                                'ops.enable_deferred_wgrad(False)' % (_deferred.get('owner'), owner))


def _wgrad_partials_for(w, d):
  """(buffer, splits) when this block's dw reduction can be deferred, else (None, 1)."""
  info = getattr(d, '_wg_split', None)
  tune = lib().ms_tuning_epoch()
  if info is None or info[2] != tune:
    sp = ctypes.c_int(1)
    n = lib().ms_wgrad_partials_elems(ctypes.byref(d), ctypes.byref(sp))
    info = d._wg_split = (n, sp.value, tune)
  if not info[0]:
    return None, 1
  key = (w.data_ptr(), id(d))
  buf = _deferred['bufs'].get(key)
  if buf is None or buf[0].numel() != info[0]:
    buf = _deferred['bufs'][key] = (torch.empty(info[0], dtype=torch.float32, device=w.device), d,   # d: keeps id(d) unique
                                    w.untyped_storage().data_ptr())
  return buf[0], info[1]


def _flush_deferred_wgrad():
  jobs = _deferred['jobs']
  _deferred['queued'] = False
  if _deferred['launches']:
    _deferred['launches'] = 0
    try:
      check(lib().ms_wgrad_flush(_stream()), 'ms_wgrad_flush')
    finally:
      _deferred['keep'].clear()
  if not jobs:
    return
  # one launch adds every job into its gradient slot; a slot that receives more than one job (a parameter used twice in the
  # step: the discriminator on fake and real poses, the decoder on the audio and the pose branch) takes them in successive
  # launches, in the order the backward pass produced them: no two workgroups of a launch add into the same words
  rounds, seen = [], {}
  for j in jobs:
    k = seen.get(j[1].data_ptr(), 0)
    seen[j[1].data_ptr()] = k + 1
    if k == len(rounds):
      rounds.append([])
    rounds[k].append(j)
  try:
    for rj in rounds:
      n = len(rj)
      parts = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in rj])
      outs = (ctypes.c_void_p * n)(*[j[1].data_ptr() for j in rj])
      elems = (ctypes.c_int * n)(*[j[1].numel() for j in rj])
      splits = (ctypes.c_int * n)(*[j[2] for j in rj])
      check(lib().ms_wgrad_reduce_multi(n, parts, outs, elems, splits, _stream()), 'ms_wgrad_reduce_multi')
  finally:
    jobs.clear()


# ------------------------------------------------------------------------------------------------
# conv block
class ConvGeom:
  """Static geometry of one conv block (the part of ms_conv_desc that does not depend on the input)."""
  __slots__ = ('nd', 'groups', 'KH', 'KW', 'SH', 'SW', 'PH', 'PW', 'slope', 'eps', 'momentum', '_cache')

  def __init__(self, nd, groups, kernel, stride, padding, slope=0.2, eps=1e-5, momentum=0.1):
    def two(v):
      if isinstance(v, (tuple, list)):
        return (1, int(v[0])) if len(v) == 1 else (int(v[0]), int(v[1]))
      return (int(v), int(v)) if nd == 2 else (1, int(v))
    self.nd, self.groups = nd, groups
    self.KH, self.KW = two(kernel)
    self.SH, self.SW = two(stride)
    ph, pw = two(padding)
    self.PH, self.PW = (ph, pw) if nd == 2 else (0, pw)
    if nd == 1:
      self.KH, self.SH = 1, 1
    self.slope, self.eps, self.momentum = float(slope), float(eps), float(momentum)
    self._cache = {}

  def desc(self, B, Cin_g, H, W, Cout_g, mode, in_mode, dtype=0):
    key = (B, Cin_g, H, W, Cout_g, mode, in_mode, dtype)
    d = self._cache.get(key)
    if d is None:
      OH = (H + 2 * self.PH - self.KH) // self.SH + 1
      OW = (W + 2 * self.PW - self.KW) // self.SW + 1
      if OH < 1 or OW < 1:
        raise RuntimeError('conv block: input (%d,%d) too small for kernel (%d,%d)' % (H, W, self.KH, self.KW))
      d = ConvDesc(B, Cin_g, H, W, Cout_g, self.groups, self.KH, self.KW, self.SH, self.SW, self.PH, self.PW,
                   OH, OW, mode, in_mode, self.slope, self.eps, self.momentum, dtype)
      d._tune = -1
      self._cache[key] = d
    L = lib()
    tune = L.ms_tuning_epoch()
    if d._tune != tune:          # scratch needs follow the kernel choice (precision mode, test knobs)
      d._fwd_ws = L.ms_conv_block_fwd_workspace(ctypes.byref(d))
      d._bwd_ws = L.ms_conv_block_bwd_workspace(ctypes.byref(d))
      d._tune = tune
    return d


# Two passes of a module side by side in one batch (gan.py:120,126: the discriminator on the fake, then on the real poses): inside
# `stat_pair()` every conv block's descriptor carries MS_DT_STAT_PAIR -- BN_TRAIN blocks take their batch statistics per half of the
# batch and move the running statistics twice, first half first (include/mixstage.h).
_stat_pair = {'on': False}


@contextlib.contextmanager
def stat_pair():
  old = _stat_pair['on']
  _stat_pair['on'] = True
  try:
    yield
  finally:
    _stat_pair['on'] = old


def stat_pair_active():
  return _stat_pair['on']


def stat_pair_ok(geom, B2, Cin_g, W, Cout_g, mode, dtype=0):
  """Can the 1-D block of this geometry run a pair of passes (a batch of B2 = 2 B clips) -> (ok, output width)."""
  d = geom.desc(B2, Cin_g, 1, W, Cout_g, mode, MS_IN_PLAIN, dtype | MS_DT_STAT_PAIR)
  return bool(lib().ms_stat_pair_ok(ctypes.byref(d))), d.OW


LATE = 'late'      # _grad_slot: truthy (autograd gets None for this gradient) but not the slot itself


def _grad_slot(param, shape_like):
  """Direct gradient write-through: FlatAdam gives every parameter a view into the flat gradient buffer
  (`_ms_grad_slot`) and marks it fresh at zero_grad.  The first gradient of a step is written straight into the slot
  by the kernels (autograd then gets None for it: no accumulate kernel); later contributions in the same step go to a
  temporary that the deferred reduction adds (second value LATE), or -- without the deferred reductions -- that autograd adds."""
  slot = getattr(param, '_ms_grad_slot', None) if param is not None else None
  if slot is not None and torch.is_grad_enabled() is False:
    if getattr(param, '_ms_grad_fresh', False):
      param._ms_grad_fresh = False
      return slot, True
    if _deferred['on']:
      # a later contribution of the same step (the discriminator runs on the fake and on the real poses): the kernels write a
      # temporary, and the ONE reduction launch at the end of the backward pass adds it into the slot (autograd gets None:
      # no accumulate kernel per parameter)
      tmp = torch.empty_like(slot)
      _deferred['jobs'].append((tmp, slot, 1))
      _queue_deferred_flush()
      return tmp, LATE
  return torch.empty_like(shape_like), False


_link_stats = {'in_launch': 0}       # data-gradient launches that took a residual gradient (tests)


class ResidualLink:
  """A tensor with TWO consumers inside one module whose gradients this package produces itself -- the UNet's down-path outputs
  (layers.py:139-151 of the reference: input of the next down block AND residual of the up path).  Autograd would add the two
  gradients with a launch of its own per level; instead the up-path block (role 'residual': it always runs first in the backward
  pass, every deeper block depends on it) leaves its residual gradient here and reports None, and the down-path block (role
  'consumer') hands it to its data-gradient launch as ms_bwd_options.dx_accum: dx leaves as the complete gradient.  `armed` is set
  by the consumer's forward (it runs first) when its launch has that form; otherwise both ends behave as if the link were absent.
  `taken`: the consumer's backward has run -- a residual gradient that arrives after it (an order the data dependencies of the UNet
  rule out, kept safe anyway) goes to autograd like any other."""
  __slots__ = ('armed', 'grad', 'taken')

  def __init__(self):
    self.armed, self.grad, self.taken = False, None, False


class _ConvBlockFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, x2, w, bias, gamma, beta, geom, mode, in_mode, stats, pre=None, prev=None, link=None):
    rm, rv = stats if stats is not None else (None, None)
    _need_hip(x, x2, w, bias, gamma, beta, rm, rv)
    x = x.contiguous()
    x2 = x2.contiguous() if x2 is not None else None
    nd = geom.nd
    if x.dim() != nd + 2:
      raise RuntimeError('conv block expects a %d-D input, got %s' % (nd + 2, tuple(x.shape)))
    B = x.shape[0]
    ref = x2 if in_mode == MS_IN_UP2ADD else x
    H, W = (ref.shape[2], ref.shape[3]) if nd == 2 else (1, ref.shape[2])
    ctot = w.shape[0]
    Cout_g = ctot // geom.groups
    Cin_g = w.shape[1]
    exp_c = Cin_g if in_mode == MS_IN_BCAST else Cin_g * geom.groups
    if x.shape[1] != exp_c:
      raise RuntimeError('conv block: expected %d input channels, got %d' % (exp_c, x.shape[1]))
    if in_mode == MS_IN_UP2ADD and (x.shape[2] * 2 != W or x2.shape[1] != exp_c):
      raise RuntimeError('UP2ADD: a %s and residual %s do not match' % (tuple(x.shape), tuple(x2.shape)))
    pair = MS_DT_STAT_PAIR if (_stat_pair['on'] and pre is None) else 0
    d = geom.desc(B, Cin_g, H, W, Cout_g, mode, in_mode, pair)
    oshape = (B, ctot, d.OH, d.OW) if nd == 2 else (B, ctot, d.OW)
    if pre is not None:
      # the block's results were produced by a launch that chains several blocks (decoder_chain): nothing to run here, this
      # node only carries the block's backward pass
      y_raw, y, save = pre
      if tuple(y.shape) != oshape:
        raise RuntimeError('precomputed block output has shape %s, expected %s' % (tuple(y.shape), oshape))
    else:
      y = torch.empty(oshape, dtype=torch.float32, device=x.device)
      y_raw = save = None
      if mode == MS_BN_TRAIN:
        y_raw = torch.empty_like(y)
        save = torch.empty((8 if pair else 4) * ctot, dtype=torch.float32, device=x.device)     # (one vector per statistics group)
    ws = workspace(d._fwd_ws, x.device) if pre is None else None
    planes = _prepared_for(w, d, 'fwd') if pre is None else None
    if pre is None:
      # BN_TRAIN blocks may finish inside the conv launch (clip-resident kernels): their workgroups meet through these counters
      sync = None
      if mode == MS_BN_TRAIN and nd == 1:
        from . import ops16
        sync = ops16.block_sync(x.device, d)
      opt = FwdOptions(planes.data_ptr() if planes is not None else None, sync.data_ptr() if sync is not None else None,
                       sync.numel() if sync is not None else 0)
      check(lib().ms_conv_block_fwd_ex(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta),
                                       _ptr(rm), _ptr(rv), _ptr(y_raw), _ptr(y), _ptr(save), _ptr(ws), ws.numel(),
                                       _stream(), ctypes.byref(opt)), 'ms_conv_block_fwd_ex')
    ctx.geom_desc = d
    ctx.mode, ctx.in_mode = mode, in_mode
    # `prev`: the autograd node of the BN_TRAIN block that produced x, handed over by a container that knows x has no other
    # consumer (layers.py: the 1-D stacks).  If this block's data-gradient launch can carry that block's BatchNorm + LeakyReLU
    # backward (ms_bwd_options.prev_*), the backward pass fuses the two: one launch and one pass over dy fewer per block.
    ctx.prev = None
    ctx.dy_is_dyr = False
    if prev is not None and pre is None and in_mode == MS_IN_PLAIN and nd == 1 and getattr(prev, 'mode', None) == MS_BN_TRAIN \
        and not bn_sync_active() and lib().ms_dgrad_fuses_prev_bn(ctypes.byref(d)):
      from . import ops16
      py = prev.saved_tensors[5]
      # (x must be an interior value nobody watches: a tensor hook or retain_grad() on it wants the gradient of the producer's
      # OUTPUT, which the fused launch never forms -- it hands the producer dy_raw)
      watched = bool(getattr(x, '_backward_hooks', None)) or bool(getattr(x, 'retains_grad', False))
      if ops16.in_launch_meetings() and not watched and py is not None and py.data_ptr() == x.data_ptr() and tuple(py.shape) == tuple(x.shape):
        ctx.prev = prev
    # `link`: (ResidualLink, role) -- see ResidualLink
    ctx.link = None
    if link is not None and pre is None and nd == 1 and not bn_sync_active():
      lk, role = link
      if role == 'consumer' and in_mode == MS_IN_PLAIN and x.requires_grad and lib().ms_dgrad_takes_accum(ctypes.byref(d)) \
          and not (getattr(x, '_backward_hooks', None) or getattr(x, 'retains_grad', False)):
        lk.armed = True
        ctx.link = link
      elif role == 'residual' and in_mode == MS_IN_UP2ADD and lk.armed:
        ctx.link = link
    ctx.has_bias = bias is not None
    ctx.params = (w, bias, gamma, beta)          # parameter objects (for their gradient slots)
    # BN_TRAIN: the block's output is saved too -- the one-launch BatchNorm backward takes x_hat and the activation mask from it
    # wherever the map inverts safely, and the in-launch forward forms (chained decoder, clip-resident blocks) write y_raw only
    # for the other channels (include/mixstage.h: ms_fwd_options.bn_sync)
    ctx.save_for_backward(x, x2, w, gamma, y_raw, y if mode in (MS_LRELU, MS_BN_TRAIN) else None, save)
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
    is_dyr = bool(ctx.dy_is_dyr)          # a consumer's fused launch already ran this block's BatchNorm + activation backward
    ctx.dy_is_dyr = False                 # (one backward pass: the consumer sets it again if the graph is walked once more)
    dyr = torch.empty_like(dy) if (mode != MS_BARE and not is_dyr) else None
    up2 = in_mode == MS_IN_UP2ADD
    want_dx = need_x or (up2 and need_x2)
    dx = torch.empty_like(x) if want_dx else None
    dx2 = torch.empty_like(x2) if (want_dx and up2) else None
    pw, pbias, pgamma, pbeta = ctx.params
    dw = dbias = dgamma = dbeta = None
    direct_w = direct_b = direct_g = direct_be = False
    if need_w:
      dw, direct_w = _grad_slot(pw, w)
      if ctx.has_bias and not is_dyr:
        dbias, direct_b = _grad_slot(pbias, pbias)
    if need_bn and not is_dyr:
      dgamma, direct_g = _grad_slot(pgamma, gamma)
      dbeta, direct_be = _grad_slot(pbeta, gamma)
    if is_dyr:
      direct_b = direct_g = direct_be = True        # (written by the consumer's launch into the slots: autograd gets None)
    # ---- fuse the producer's BatchNorm backward into this block's data gradient?
    fuse = None
    prev = ctx.prev
    if prev is not None and want_dx and not up2 and _overlap['stream'] is None and not torch.is_grad_enabled():
      ppw, ppbias, ppgamma, ppbeta = prev.params
      p_need_w, p_need_bn = prev.needs_input_grad[2], prev.needs_input_grad[4]
      wanted = ([ppbias] if (p_need_w and prev.has_bias) else []) + ([ppgamma, ppbeta] if p_need_bn else [])
      if all(getattr(q, '_ms_grad_slot', None) is not None and (getattr(q, '_ms_grad_fresh', False) or _deferred['on']) for q in wanted):
        _, _, _, pgam_t, py_raw, py, psave = prev.saved_tensors
        if py_raw is not None and py is not None and psave is not None:
          from . import ops16
          sync = ops16.block_sync(dev, d, tag='dgrad_bn')     # (its own counters: the launch's member count is this block's pixel workgroups)
          if sync is not None:                # (None: in-launch meetings were switched off after the forward pass -- no fusion, and no slot taken yet)
            pdb = _grad_slot(ppbias, ppbias)[0] if (p_need_w and prev.has_bias) else None
            pdg = _grad_slot(ppgamma, pgam_t)[0] if p_need_bn else None
            pdbe = _grad_slot(ppbeta, pgam_t)[0] if p_need_bn else None
            fuse = (py, py_raw, psave, pgam_t, pdg, pdbe, pdb, sync, float(prev.geom_desc.slope))
    # ---- the gradient the input's other consumer left for this launch (ResidualLink)
    acc = None
    if ctx.link is not None and ctx.link[1] == 'consumer':
      acc, ctx.link[0].grad = ctx.link[0].grad, None
      ctx.link[0].taken = True
      if acc is not None and (not want_dx or tuple(acc.shape) != tuple(x.shape)):
        raise RuntimeError('residual link: a residual gradient of shape %s arrived for an input of shape %s (needs grad: %s)' %
                           (tuple(acc.shape), tuple(x.shape), want_dx))
    acc_in_launch = False
    ws = workspace(d._bwd_ws, dev)
    side = _overlap['stream']
    if side is not None and need_w and direct_w is not True:
      # a second contribution to a parameter in this step (autograd will add it on this stream): the first one may
      # still be in flight on the side stream
      torch.cuda.current_stream().wait_stream(side)
    if side is not None and direct_w is True and (dbias is None or direct_b is True):
      # dw goes straight into the flat gradient buffer, so nothing downstream in autograd reads it: run it on the side
      # stream.  Its inputs must outlive this function until the streams are joined.
      ws2 = side_workspace(d._bwd_ws, dev)
      _overlap['keep'].append((x, x2, dy, dyr))
      check(lib().ms_conv_block_bwd_overlap(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(gamma), None, None,
                                            _ptr(y_raw), _ptr(y), _ptr(save), _ptr(dy), _ptr(dyr), _ptr(dx), _ptr(dx2),
                                            _ptr(dw), _ptr(dbias), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(),
                                            _stream(), _vp(side.cuda_stream), _ptr(ws2), ws2.numel()),
            'ms_conv_block_bwd_overlap')
    else:
      wt = _prepared_for(w, d) if want_dx else None
      part, nsplit = (None, 1)
      if _deferred['on'] and direct_w is True:
        part, nsplit = _wgrad_partials_for(w, d)
      # the weight-gradient kernel itself is queued (ms_wgrad_flush at the end of the backward pass) when nothing downstream
      # reads dw: it lands in the flat gradient buffer
      defer_launch = bool(_deferred['on'] and direct_w is True and dw is not None and DEFER_WGRAD_LAUNCH)
      if wt is not None or part is not None or defer_launch or fuse is not None or is_dyr or acc is not None:
        opt = BwdOptions(None, None, 0, wt.data_ptr() if wt is not None else None,
                         part.data_ptr() if part is not None else None, 1 if defer_launch else 0)
        if acc is not None:
          acc = acc.contiguous()
          opt.dx_accum = acc.data_ptr()
          acc_in_launch = True
          _link_stats['in_launch'] += 1
        if fuse is not None:
          py, py_raw, psave, pgam_t, pdg, pdbe, pdb, sync, pslope = fuse
          opt.prev_y, opt.prev_y_raw, opt.prev_save, opt.prev_gamma = py.data_ptr(), py_raw.data_ptr(), psave.data_ptr(), pgam_t.data_ptr()
          opt.prev_dgamma = pdg.data_ptr() if pdg is not None else None
          opt.prev_dbeta = pdbe.data_ptr() if pdbe is not None else None
          opt.prev_dbias = pdb.data_ptr() if pdb is not None else None
          opt.prev_slope, opt.bn_sync, opt.bn_sync_words = pslope, sync.data_ptr(), sync.numel()
          prev.dy_is_dyr = True            # the producer's node runs after this one: its incoming gradient is dy_raw already
        opt.dy_is_dyr = 1 if is_dyr else 0
        check(lib().ms_conv_block_bwd_ex(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(gamma), None, None, _ptr(y_raw),
                                         _ptr(y), _ptr(save), _ptr(dy), _ptr(dyr), _ptr(dx), _ptr(dx2), _ptr(dw),
                                         _ptr(dbias), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), _stream(),
                                         ctypes.byref(opt)), 'ms_conv_block_bwd_ex')
        if defer_launch:
          _deferred['launches'] += 1
          _deferred['keep'].append((x, x2, dy, dyr, dw, part))
          _queue_deferred_flush()
        if part is not None:
          _deferred['jobs'].append((part, dw, nsplit))
          _queue_deferred_flush()
      else:
        check(lib().ms_conv_block_bwd(ctypes.byref(d), _ptr(x), _ptr(x2), _ptr(w), _ptr(gamma), None, None, _ptr(y_raw),
                                      _ptr(y), _ptr(save), _ptr(dy), _ptr(dyr), _ptr(dx), _ptr(dx2), _ptr(dw),
                                      _ptr(dbias), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), _stream()),
              'ms_conv_block_bwd')
    if acc is not None and not acc_in_launch:
      dx.add_(acc)                          # (side-stream experiment path: no options struct)
    if ctx.link is not None and ctx.link[1] == 'residual' and dx2 is not None and ctx.needs_input_grad[1] and not ctx.link[0].taken:
      ctx.link[0].grad = dx2                # the consumer's data-gradient launch adds it: autograd gets None for the residual
      dx2 = None
    return (dx, dx2, None if direct_w else dw, None if direct_b else dbias, None if direct_g else dgamma,
            None if direct_be else dbeta, None, None, None, None, None, None, None)


def _f32(t):
  return t.float() if isinstance(t, torch.Tensor) and t.dtype == torch.float64 else t


def _bridge64(fn):
  """float64 boundary (the reference trainer casts the model with .double(), trainer.py:138, and feeds float64 batches,
  dataUtils.py:547): the kernels compute in fp32, so float64 tensors are cast on the way in (a differentiable cast: gradients
  arrive back in float64) and fp32 results on the way out.  The state_dict keeps its float64 tensors."""
  def wrapped(*args, **kw):
    vals = list(args) + list(kw.values())
    if not any(isinstance(a, torch.Tensor) and a.dtype == torch.float64 for a in vals):
      return fn(*args, **kw)
    out = fn(*[_f32(a) for a in args], **{k: _f32(v) for k, v in kw.items()})
    back = lambda o: o.double() if isinstance(o, torch.Tensor) and o.dtype == torch.float32 else o
    return tuple(back(o) for o in out) if isinstance(out, tuple) else back(out)
  wrapped.__doc__, wrapped.__name__ = fn.__doc__, fn.__name__
  return wrapped


# MS_CHAIN_BN=0 / enable_chain_fusion(False): the 1-D stacks keep every block's BatchNorm backward in its own launch (ablations;
# the side-stream weight-gradient experiment, whose launches have no fused form)
_chain_fusion = {'on': os.environ.get('MS_CHAIN_BN', '1') != '0'}


def enable_chain_fusion(on):
  old = _chain_fusion['on']
  _chain_fusion['on'] = bool(on)
  return old


def conv_block(x, w, bias, geom, mode, gamma=None, beta=None, running_mean=None, running_var=None, x2=None,
               in_mode=MS_IN_PLAIN, chain_prev=False, link=None):
  """One conv block of the path on the HIP kernels (see include/mixstage.h: ms_conv_block_fwd/bwd)."""
  if w.dtype == torch.float64 or x.dtype == torch.float64:
    # .double() model: fp32 shadows of the parameters (differentiable casts) and of the running statistics, which the
    # kernels update in place and which are written back to the float64 buffers
    rm32, rv32 = _f32(running_mean), _f32(running_var)
    stats = (rm32, rv32) if rm32 is not None else None
    y = _ConvBlockFn.apply(_f32(x), _f32(x2), _f32(w), _f32(bias), _f32(gamma), _f32(beta), geom, mode, in_mode, stats)
    if mode == MS_BN_TRAIN and running_mean is not None and running_mean.dtype == torch.float64:
      with torch.no_grad():
        running_mean.copy_(rm32)
        running_var.copy_(rv32)
    return y.double()
  stats = (running_mean, running_var) if running_mean is not None else None
  # chain_prev: the caller vouches that x -- the output of another conv block -- feeds nothing but this block
  prev = x.grad_fn if (chain_prev and _chain_fusion['on'] and x.grad_fn is not None and type(x.grad_fn).__name__ == '_ConvBlockFnBackward') else None
  return _ConvBlockFn.apply(x, x2, w, bias, gamma, beta, geom, mode, in_mode, stats, None, prev,
                            link if (link is not None and _chain_fusion['on'] and torch.is_grad_enabled()) else None)


# ------------------------------------------------------------------------------------------------
# Backward-pass marker: an identity in the forward pass whose backward runs a callback -- the point of the backward pass at
# which every gradient of the layers BEHIND it in the forward pass (the grouped decoder, logits, cluster classifier) is
# complete.  A data-parallel trainer starts the gradient exchange of that bucket there, next to the rest of the backward pass
# (train_step.MixStageTrainStep); without a callback the marker is not inserted at all.
_marker = {'cb': None}


def set_backward_marker(callback):
  _marker['cb'] = callback


class _BackwardMarkerFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return x.view_as(x)

  @staticmethod
  def backward(ctx, g):
    cb = _marker['cb']
    if cb is not None:
      cb()
    return g


def backward_marker(x):
  return _BackwardMarkerFn.apply(x) if (_marker['cb'] is not None and x.requires_grad) else x


# ------------------------------------------------------------------------------------------------
# cross-rank ("global") BatchNorm for data-parallel training (include/mixstage.h: ms_bn_stats ...)
_bn_sync = {'group': None, 'on': False}


def set_bn_sync(on, process_group=None):
  """bn_sync='global': every BatchNorm(train) block normalises with the statistics of the GLOBAL batch (all ranks), as the
  single device of the reference does (layers.py:65-70).  Two small collectives per block and direction; not capturable in
  a HIP graph (the exchanges run between the kernels)."""
  _bn_sync['on'], _bn_sync['group'] = bool(on), process_group


def bn_sync_active():
  import torch.distributed as dist
  return _bn_sync['on'] and dist.is_available() and dist.is_initialized() and dist.get_world_size(_bn_sync['group']) > 1


class _SyncBNActFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, y_raw, gamma, beta, rm, rv, slope, eps, momentum):
    import torch.distributed as dist
    _need_hip(y_raw, gamma, beta, rm, rv)
    y_raw = y_raw.contiguous()
    B, C = y_raw.shape[0], y_raw.shape[1]
    HW = y_raw.numel() // (B * C)
    group = _bn_sync['group']
    world = dist.get_world_size(group)
    stats = torch.empty((C, 2), dtype=torch.float32, device=y_raw.device)
    check(lib().ms_bn_stats(_ptr(y_raw), _ptr(stats), B, C, HW, _stream()), 'ms_bn_stats')
    allstats = torch.empty((world, C, 2), dtype=torch.float32, device=y_raw.device)
    dist.all_gather_into_tensor(allstats, stats, group=group) if dist.get_backend(group) == 'nccl' else \
        allstats.copy_(torch.stack(_all_gather_list(stats, world, group)))
    y = torch.empty_like(y_raw)
    save = torch.empty(4 * C, dtype=torch.float32, device=y_raw.device)
    check(lib().ms_bn_train_apply(_ptr(allstats), world, B * HW, _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), _ptr(y_raw),
                                  _ptr(y), _ptr(save), B, C, HW, eps, momentum, slope, _stream()), 'ms_bn_train_apply')
    ctx.save_for_backward(y_raw, gamma, save)
    ctx.meta = (B, C, HW, slope, world, group)
    ctx.params = (gamma, beta)
    return y

  @staticmethod
  def backward(ctx, dy):
    import torch.distributed as dist
    y_raw, gamma, save = ctx.saved_tensors
    B, C, HW, slope, world, group = ctx.meta
    dy = dy.contiguous()
    sums = torch.empty((C, 2), dtype=torch.float32, device=dy.device)
    ws = workspace(lib().ms_bn_bwd_workspace(B, C), dy.device)
    check(lib().ms_bn_bwd_sums(_ptr(dy), _ptr(y_raw), _ptr(save), _ptr(sums), B, C, HW, slope, _ptr(ws), ws.numel(), _stream()),
          'ms_bn_bwd_sums')
    # this rank's share of dgamma / dbeta (the gradient all-reduce averages the shares); the normalisation needs the global sums
    pgamma, pbeta = ctx.params
    dgamma, direct_g = _grad_slot(pgamma, gamma)
    dbeta, direct_b = _grad_slot(pbeta, gamma)
    dgamma.copy_(sums[:, 1]); dbeta.copy_(sums[:, 0])
    gsum = sums.clone()
    dist.all_reduce(gsum, op=dist.ReduceOp.SUM, group=group)
    dyr = torch.empty_like(dy)
    check(lib().ms_bn_bwd_apply(_ptr(dy), _ptr(y_raw), _ptr(save), _ptr(gamma), _ptr(gsum), float(world) * B * HW, _ptr(dyr), B, C,
                                HW, slope, _stream()), 'ms_bn_bwd_apply')
    return dyr, None if direct_g else dgamma, None if direct_b else dbeta, None, None, None, None, None


def _all_gather_list(t, world, group):
  import torch.distributed as dist
  out = [torch.empty_like(t) for _ in range(world)]
  dist.all_gather(out, t, group=group)
  return out


def sync_bn_act(y_raw, gamma, beta, running_mean, running_var, slope, eps, momentum):
  """BatchNorm(train, statistics of the global batch over all ranks) + LeakyReLU of a bare conv output."""
  return _SyncBNActFn.apply(y_raw, gamma, beta, running_mean, running_var, float(slope), float(eps), float(momentum))


# ------------------------------------------------------------------------------------------------
# Chained pose decoder (include/mixstage.h: ms_decoder_chain_fwd): decoder.0-3 + logits + softmax mixture in one launch.
CHAIN_SYNC_FIRST_WORD = 32             # (word 0 of the chain's sync buffer is the error flag: ops16.chain_sync)
USE_DECODER_CHAIN = os.environ.get('MS_DECODER_CHAIN', '1') != '0'       # ablations: MS_DECODER_CHAIN=0 runs the blocks one by one


def _chain_desc(B, M, T, cin0, P, mode, blk, dtype=0):
  g = blk._geometry()
  return _lib.ChainDesc(B, M, T, cin0, 256, P, 4, mode, dtype, CHAIN_SYNC_FIRST_WORD, 0, g.slope, g.eps, g.momentum)


_chain_scratch = {}


def _chain_prepared(d, ws):
  """The chain's weight streams for these five weights.  Under a trainer that owns the parameter updates
  (enable_prepared_weights): built on first use, rebuilt when torch changed a weight (version counters) or when the trainer says
  so after an update it made through raw pointers (refresh_prepared_weights).  Otherwise: rebuilt on every call (nobody
  vouches for the weights between calls), into a buffer that is reused."""
  key = (ws[0].data_ptr(), d.M, d.cin0, d.P, d.dtype, 'chain32')
  if not _prepared['on']:
    n = (lib().ms_decoder_chain_prepared_bytes(ctypes.byref(d)) + 3) // 4
    buf = _chain_scratch.get(key)
    if buf is None or buf.numel() != n:
      _chain_scratch.clear()
      buf = _chain_scratch[key] = torch.empty(n, dtype=torch.float32, device=ws[0].device)
    _prepare_entries([dict(w=ws[0], ws=list(ws), d=d, n=n, wt=buf, version=-1, tune=-1, kind='chain32')])
    return buf
  e = _prepared['entries'].get(key)
  if e is None:
    n = (lib().ms_decoder_chain_prepared_bytes(ctypes.byref(d)) + 3) // 4
    e = dict(w=ws[0], ws=list(ws), d=d, n=n, wt=torch.empty(n, dtype=torch.float32, device=ws[0].device), version=-1, tune=-1,
             kind='chain32')
    _prepared['entries'][key] = e
    _prepared['by_storage'].setdefault(ws[0].untyped_storage().data_ptr(), []).append(e)
  e['ws'] = list(ws)
  if e['version'] != sum(x._version for x in ws):
    _prepare_entries([e])
  return e['wt']


def decoder_chain(x, blocks, logits, score, P):
  """(out (B,T,P), soft (B,T,M)) = softmax mixture of logits(decoder(x)) for the M sub-generators (JL:190-194) in ONE launch, or
  None when this shape / mode / device is not served by the chained kernel (the caller then runs the blocks one by one).
  blocks: the four ConvNormRelu modules of the grouped decoder; logits: the grouped 1x1 nn.Conv1d; x (B, cin0, T) fp32 shared by
  all groups; score (B, M, T)."""
  if not USE_DECODER_CHAIN or len(blocks) != 4 or x.dim() != 3 or not x.is_cuda or x.dtype != torch.float32 or bn_sync_active():
    return None
  from . import ops16
  if not ops16.in_launch_meetings():
    return None
  blk0 = blocks[0]
  M = blk0.conv.groups
  B, cin0, T = x.shape
  if any(getattr(m, '_ms_dt', 0) for m in blocks) or any(m._p and m.training for m in blocks):
    return None
  if any(m._forward_hooks or m._forward_pre_hooks for m in list(blocks) + [logits]):
    return None                            # (someone observes the blocks' own forward calls: run them)
  if any(m.conv.groups != M or m.conv.kernel_size != (3,) or m.conv.stride != (1,) or m.conv.padding != (1,) or
         m.conv.weight.shape[0] != 256 * M or m.conv.weight.dtype != torch.float32 or m._slope != blk0._slope for m in blocks):
    return None
  if blocks[0].conv.weight.shape[1] != cin0 or any(m.conv.weight.shape[1] != 256 for m in blocks[1:]):
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
    return None                            # (eval-mode blocks are never differentiated on the path)
  d = _chain_desc(B, M, T, cin0, P, mode, blk0)
  if not lib().ms_decoder_chain_supported(ctypes.byref(d)):
    return None
  _need_hip(x, score, *[t for t in params if t is not None])
  x, score = x.contiguous(), score.contiguous()
  dev = x.device
  ws = [m.conv.weight for m in blocks] + [logits.weight]
  prepared = _chain_prepared(d, ws)
  sync = ops16.chain_sync(dev, B, M, CHAIN_SYNC_FIRST_WORD + lib().ms_decoder_chain_sync_words(ctypes.byref(d)))
  C = 256 * M
  keep = need_grad
  y_raw = [torch.empty((B, C, T), dtype=torch.float32, device=dev) if keep else None for _ in blocks]
  y = [torch.empty((B, C, T), dtype=torch.float32, device=dev) if keep else None for _ in blocks]
  save = [torch.empty(4 * C, dtype=torch.float32, device=dev) if keep else None for _ in blocks]
  z = torch.empty((B, M * P, T), dtype=torch.float32, device=dev) if keep else None
  soft = torch.empty((B, T, M), dtype=torch.float32, device=dev)
  out = torch.empty((B, T, P), dtype=torch.float32, device=dev)
  tn = _lib.ChainTensors()
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
  # the autograd graph of the unchained forward pass, every node carrying results the chain already produced: the backward
  # pass is the blocks' own (ms_conv_block_bwd, ms_softmax_mix_bwd)
  h = x
  for l, m in enumerate(blocks):
    n = m.norm
    h = _ConvBlockFn.apply(h, None, m.conv.weight, m.conv.bias, n.weight, n.bias, m._geometry(), MS_BN_TRAIN,
                           MS_IN_BCAST if l == 0 else MS_IN_PLAIN, (n.running_mean, n.running_var), (y_raw[l], y[l], save[l]))
  geom = getattr(logits, '_ms_geom', None)
  if geom is None:
    geom = logits._ms_geom = ConvGeom(1, logits.groups, logits.kernel_size, logits.stride, logits.padding, slope=0.0)
  zt = _ConvBlockFn.apply(h, None, logits.weight, logits.bias, None, None, geom, MS_BARE, MS_IN_PLAIN, None, (None, z, None))
  return _SoftmaxMixFn.apply(zt, score, int(P), (out, soft))


# ------------------------------------------------------------------------------------------------
class _LerpTimeFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, t_out):
    _need_hip(x)
    x = x.contiguous()
    B, C, Tin, F = x.shape
    y = torch.empty((B, C, t_out), dtype=torch.float32, device=x.device)
    check(lib().ms_lerp_time_fwd(_ptr(x), _ptr(y), B, C, Tin, F, t_out, _stream()), 'ms_lerp_time_fwd')
    ctx.shape = (B, C, Tin, F, t_out)
    return y

  @staticmethod
  def backward(ctx, dy):
    B, C, Tin, F, t_out = ctx.shape
    dy = dy.contiguous()
    dx = torch.empty((B, C, Tin, F), dtype=torch.float32, device=dy.device)
    check(lib().ms_lerp_time_bwd(_ptr(dy), _ptr(dx), B, C, Tin, F, t_out, _stream()), 'ms_lerp_time_bwd')
    return dx, None


@_bridge64
def lerp_time(x, t_out):
  """F.interpolate(x, size=(t_out, 1), mode='bilinear').squeeze(-1) (layers.py:197-198)."""
  return _LerpTimeFn.apply(x, int(t_out))


class _SoftmaxMixFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, z, score, P, pre=None):
    _need_hip(z, score)
    z, score = z.contiguous(), score.contiguous()
    B, M, T = score.shape
    assert z.shape == (B, M * P, T), (z.shape, score.shape, P)
    if pre is not None:
      out, soft = pre                     # produced by the chained decoder launch (decoder_chain)
    else:
      soft = torch.empty((B, T, M), dtype=torch.float32, device=z.device)
      out = torch.empty((B, T, P), dtype=torch.float32, device=z.device)
      check(lib().ms_softmax_mix_fwd(_ptr(z), _ptr(score), _ptr(soft), _ptr(out), B, M, P, T, _stream()),
            'ms_softmax_mix_fwd')
    ctx.save_for_backward(z, soft)
    ctx.dims = (B, M, P, T)
    ctx.mark_non_differentiable(soft)
    return out, soft

  @staticmethod
  def backward(ctx, dout, _dsoft):
    z, soft = ctx.saved_tensors
    B, M, P, T = ctx.dims
    dout = dout.contiguous()
    dz = torch.empty_like(z)
    dscore = torch.empty((B, M, T), dtype=torch.float32, device=z.device)
    check(lib().ms_softmax_mix_bwd(_ptr(z), _ptr(soft), _ptr(dout), _ptr(dz), _ptr(dscore), B, M, P, T, _stream()),
          'ms_softmax_mix_bwd')
    return dz, dscore, None, None


@_bridge64
def softmax_mix(z, score, P):
  """(out (B,T,P), softmax (B,T,M)) of JL:186-187,194 from channel-major z (B,M*P,T), score (B,M,T).
  The returned softmax is a detached monitor (the reference's labels_cap_soft is only read by the
  trainer for histograms); the gradient into `score` flows through `out`."""
  return _SoftmaxMixFn.apply(z, score, int(P))


class _ConcatStyleFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, emb, ids):
    _need_hip(x, emb, ids)
    x, emb = x.contiguous(), emb.contiguous()
    if ids.dtype != torch.int64 or ids.dim() != 2:
      raise TypeError('style ids must be an int64 (B,T) tensor (expanded views are fine)')
    B, C, T = x.shape
    S, D = emb.shape
    assert ids.shape == (B, T)
    sb, st = ids.stride()
    out = torch.empty((B, C + D, T), dtype=torch.float32, device=x.device)
    check(lib().ms_concat_style_fwd(_ptr(x), _ptr(emb), _ptr(ids), sb, st, _ptr(out), B, C, D, T, _stream()),
          'ms_concat_style_fwd')
    ctx.save_for_backward(ids)
    ctx.dims = (B, C, D, T, S, sb, st)
    ctx.emb_param = emb
    return out

  @staticmethod
  def backward(ctx, dout):
    (ids,) = ctx.saved_tensors
    B, C, D, T, S, sb, st = ctx.dims
    dout = dout.contiguous()
    dx = torch.empty((B, C, T), dtype=torch.float32, device=dout.device) if ctx.needs_input_grad[0] else None
    demb, direct = (None, False)
    if ctx.needs_input_grad[1]:
      demb, direct = _grad_slot(ctx.emb_param, ctx.emb_param)
    check(lib().ms_concat_style_bwd(_ptr(dout), _ptr(ids), sb, st, _ptr(dx), _ptr(demb), B, C, D, T, S, _stream()),
          'ms_concat_style_bwd')
    return dx, None if direct else demb, None


@_bridge64
def concat_style(x, emb_weight, ids):
  """(B, C+D, T) = [x ; emb_weight[ids]^T]: EmbLin 'emb' lookup + cat of JL:175-180, channel-major."""
  return _ConcatStyleFn.apply(x, emb_weight, ids)


def _loss_scale(scale):
  """struct ms_loss_scale for a host constant or a one-float device tensor."""
  if torch.is_tensor(scale):
    if not scale.is_cuda or scale.dtype != torch.float32 or scale.numel() != 1:
      raise TypeError('a tensor loss weight is one float32 on the device')
    return _lib.LossScale(1.0, scale.data_ptr())
  return _lib.LossScale(float(scale), None)


class _CrossEntropyFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, score, target, layout, scale):
    _need_hip(score, target)
    score = score.contiguous()
    target = target.contiguous()
    if target.dtype != torch.int64:
      raise TypeError('cross entropy targets must be int64')
    if layout == 'bct':      # score (B, C, T), rows = (b, t)
      B, C, T = score.shape
      dims = (B, T, C, C * T, T, 1)
    else:                    # score (N, C)
      N, C = score.shape
      dims = (N, 1, C, C, 1, 1)
    assert target.numel() == dims[0] * dims[1]
    loss = torch.empty((), dtype=torch.float32, device=score.device)
    ls = _loss_scale(scale)          # the weight is applied inside the kernels (bit-identical to `loss * scale`, one launch less)
    check(lib().ms_cross_entropy_fwd_ex(_ptr(score), _ptr(target), _ptr(loss), *dims, _stream(), ctypes.byref(ls)),
          'ms_cross_entropy_fwd_ex')
    ctx.save_for_backward(score, target)
    ctx.dims, ctx.scale = dims, scale
    return loss

  @staticmethod
  def backward(ctx, g):
    score, target = ctx.saved_tensors
    g = g.contiguous()
    dscore = torch.empty_like(score)
    ls = _loss_scale(ctx.scale)
    check(lib().ms_cross_entropy_bwd_ex(_ptr(score), _ptr(target), _ptr(g), _ptr(dscore), *ctx.dims, 0, _stream(),
                                        ctypes.byref(ls)), 'ms_cross_entropy_bwd_ex')
    return dscore, None, None, None


@_bridge64
def cross_entropy(score, target, layout='nc', scale=1.0):
  """scale * F.cross_entropy(...) with mean reduction; layout 'bct' = class axis 1 of (B,C,T)."""
  return _CrossEntropyFn.apply(score, target, layout, float(scale))


class _VelocityFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    _need_hip(x)
    x = x.contiguous()
    B, T, P = x.shape
    v = torch.empty((B, P, T), dtype=torch.float32, device=x.device)
    check(lib().ms_velocity_fwd(_ptr(x), _ptr(v), B, T, P, _stream()), 'ms_velocity_fwd')
    ctx.dims = (B, T, P)
    return v

  @staticmethod
  def backward(ctx, dv):
    B, T, P = ctx.dims
    dv = dv.contiguous()
    dx = torch.empty((B, T, P), dtype=torch.float32, device=dv.device)
    check(lib().ms_velocity_bwd(_ptr(dv), _ptr(dx), B, T, P, _stream()), 'ms_velocity_bwd')
    return dx


@_bridge64
def velocity_cm(x):
  """GAN.get_velocity (gan.py:47-52) of x (B,T,P), returned channel-major (B,P,T) for D's first conv."""
  return _VelocityFn.apply(x)


class _TransposeFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, to_cm):
    _need_hip(x)
    x = x.contiguous()
    B, R, C = x.shape
    y = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
    fn = lib().ms_transpose_btc if to_cm else lib().ms_transpose_bct
    check(fn(_ptr(x), _ptr(y), B, R, C, _stream()), 'ms_transpose')
    ctx.to_cm = to_cm
    return y

  @staticmethod
  def backward(ctx, dy):
    return _TransposeFn.apply(dy, not ctx.to_cm), None


class _SplitHalvesFn(torch.autograd.Function):
  """x (2B, ...) -> (x[:B], x[B:]) as views; the backward pass joins the two gradients with ONE copy launch (two slice nodes would
  each fill a zero tensor of the whole batch and add)."""

  @staticmethod
  def forward(ctx, x):
    h = x.shape[0] // 2
    return x[:h], x[h:]

  @staticmethod
  def backward(ctx, g0, g1):
    return torch.cat([g0, g1], dim=0)


def split_halves(x):
  return _SplitHalvesFn.apply(x)


@_bridge64
def to_channel_major(x):
  """(B,T,C) -> contiguous (B,C,T)."""
  return _TransposeFn.apply(x, True)


@_bridge64
def to_time_major(x):
  """(B,C,T) -> contiguous (B,T,C)."""
  return _TransposeFn.apply(x, False)


class _L1MeanFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, a, b, target, scale, squared=False):
    _need_hip(a, b)
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    n = a.numel()
    loss = torch.empty((), dtype=torch.float32, device=a.device)
    part = torch.empty(lib().ms_reduce_partials_count(n), dtype=torch.float32, device=a.device)
    # `scale`: a host constant, or a device-resident weight (gan.py: lambda schedule) read when the kernels run; either way it
    # is applied inside the kernels (bit-identical to `loss * scale` / `g * scale`, without their launches)
    ls = _loss_scale(scale)
    check(lib().ms_lp_mean_fwd_ex(1 if squared else 0, _ptr(a), _ptr(b), target, _ptr(loss), _ptr(part), n, _stream(),
                                  ctypes.byref(ls)), 'ms_lp_mean_fwd_ex')
    ctx.save_for_backward(a, b, scale if torch.is_tensor(scale) else None)
    ctx.target, ctx.scale, ctx.squared = target, (None if torch.is_tensor(scale) else scale), squared
    return loss

  @staticmethod
  def backward(ctx, g):
    a, b, scale_t = ctx.saved_tensors
    g = g.contiguous()
    da = torch.empty_like(a)
    ls = _loss_scale(scale_t if scale_t is not None else ctx.scale)
    check(lib().ms_lp_mean_bwd_ex(1 if ctx.squared else 0, _ptr(a), _ptr(b), ctx.target, _ptr(g), _ptr(da), a.numel(), _stream(),
                                  ctypes.byref(ls)), 'ms_lp_mean_bwd_ex')
    return da, None, None, None, None


@_bridge64
def l1_mean(a, b=None, target=0.0, scale=1.0):
  """scale * mean|a - b| (b a tensor without grad, or the constant `target`): gan.py:64-75 with L1Loss."""
  return _L1MeanFn.apply(a, b, float(target), scale if torch.is_tensor(scale) else float(scale), False)


@_bridge64
def l2_mean(a, b=None, target=0.0, scale=1.0):
  """scale * mean (a - b)^2: gan.py:64-75 with MSELoss, the GAN constructor's default criterion."""
  return _L1MeanFn.apply(a, b, float(target), scale if torch.is_tensor(scale) else float(scale), True)


LP_PAIR_MAX = 2048      # values per half up to which the paired criterion is one launch (include/mixstage.h: ms_lp_mean_pair_fwd)


class _LpPairFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, a, t0, t1, s0, s1, squared):
    _need_hip(a)
    a = a.contiguous()
    n = a.numel() // 2
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    ls = (_lib.LossScale * 2)(_loss_scale(s0), _loss_scale(s1))
    tg = (ctypes.c_float * 2)(t0, t1)
    check(lib().ms_lp_mean_pair_fwd(1 if squared else 0, _ptr(a), tg, _ptr(out), n, _stream(), ls), 'ms_lp_mean_pair_fwd')
    ctx.save_for_backward(a, s0 if torch.is_tensor(s0) else None, s1 if torch.is_tensor(s1) else None)
    ctx.consts = (t0, t1, None if torch.is_tensor(s0) else s0, None if torch.is_tensor(s1) else s1, squared)
    return out[0], out[1]

  @staticmethod
  def backward(ctx, g0, g1):
    a, s0t, s1t = ctx.saved_tensors
    t0, t1, s0, s1, squared = ctx.consts
    da = torch.empty_like(a)
    ls = (_lib.LossScale * 2)(_loss_scale(s0t if s0t is not None else s0), _loss_scale(s1t if s1t is not None else s1))
    tg = (ctypes.c_float * 2)(t0, t1)
    g0 = g0.contiguous() if g0 is not None else None
    g1 = g1.contiguous() if g1 is not None else None
    check(lib().ms_lp_mean_pair_bwd(1 if squared else 0, _ptr(a), tg, _ptr(g0), _ptr(g1), _ptr(da), a.numel() // 2, _stream(), ls),
          'ms_lp_mean_pair_bwd')
    return da, None, None, None, None, None


def lp_mean_pair(a, targets, scales, squared=False):
  """(scales[0] * mean|a[:B] - targets[0]|, scales[1] * mean|a[B:] - targets[1]|) (squared: the MSELoss form) of the two halves of a
  fp32 tensor in one launch each way -- the same bits as two l1_mean / l2_mean calls on the halves (gan.py:121,127 on the paired
  discriminator pass).  At most LP_PAIR_MAX values per half."""
  s0, s1 = [s if torch.is_tensor(s) else float(s) for s in scales]
  return _LpPairFn.apply(a, float(targets[0]), float(targets[1]), s0, s1, bool(squared))


def copy_multi(pairs):
  """dst.copy_(src) for every (dst, src) pair of contiguous device tensors of equal size and dtype, in ONE launch."""
  pairs = [(d, s) for d, s in pairs if d.numel()]
  if not pairs:
    return
  for d, s in pairs:
    if not (d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype and d.shape == s.shape):
      raise TypeError('copy_multi: contiguous device tensors of equal shape and dtype')
  n = len(pairs)
  srcs = (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in pairs])
  dsts = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
  sizes = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in pairs])
  check(lib().ms_copy_multi(n, srcs, dsts, sizes, _stream()), 'ms_copy_multi')


# ------------------------------------------------------------------------------------------------
def grad_norm(flat_grad, out, partials):
  check(lib().ms_sqnorm(_ptr(flat_grad), flat_grad.numel(), _ptr(out), _ptr(partials), _stream()), 'ms_sqnorm')


def adam_step(p, g, m, v, norm, max_norm, lr, beta1, beta2, eps, step_state):
  check(lib().ms_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(norm), max_norm, lr, beta1, beta2,
                           eps, _ptr(step_state), _stream()), 'ms_adam_step')


def adam_step_segmented(p, g, m, v, norm, max_norm, lr, beta1, beta2, eps, step_state, seg_of_chunk, seg_first, seg_scratch):
  check(lib().ms_adam_step_segmented(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(norm), max_norm, lr, beta1, beta2,
                                     eps, _ptr(step_state), _ptr(seg_of_chunk), _ptr(seg_first), _ptr(seg_scratch),
                                     seg_first.numel(), _stream()), 'ms_adam_step_segmented')


def selftest_mfma(A, B):
  _need_hip(A, B)
  K = A.shape[1]
  C = torch.empty((32, 32), dtype=torch.float32, device=A.device)
  check(lib().ms_selftest_mfma(_ptr(A.contiguous()), _ptr(B.contiguous()), _ptr(C), K, _stream()), 'ms_selftest_mfma')
  return C


def timing_enable(on):
  """Bracket every conv / BN launch with HIP events on its stream (bench.py's live kernel timing)."""
  check(lib().ms_timing_enable(1 if on else 0), 'ms_timing_enable')


def timing_report():
  need = lib().ms_timing_report(None, 0)
  buf = ctypes.create_string_buffer(need + 16)
  lib().ms_timing_report(buf, need + 16)
  rows = []
  for line in buf.value.decode().splitlines():
    label, count, ms, flops, nbytes = line.split('\t')
    rows.append(dict(label=label, count=int(count), total_ms=float(ms), flops=float(flops), bytes=float(nbytes)))
  return rows
