"""The trainer-step contract of the reference, MI355X-native.

One step == what src/model/trainer.py does per batch (TrainerLateClusterStyleGAN):
  zero_grad (TR:604,1104-1107) -> model(x, y, **kwargs) (TR:1158-1165) -> loss = sum(internal_losses)
  (TR:1268-1285) -> loss.backward(); clip_grad_norm_(G or D params, 1); G_optim|D_optim.step() by
  model.G_flag (TR:1138-1146), with torch.optim.Adam(lr=1e-4) per network (TR:262-287, ARGS:180-183).

MI355X-first structure:
  * parameters, gradients and Adam moments of each network live in ONE flat fp32 HBM buffer each
    (FlatAdam): zero_grad is one memset, the global grad-norm one reduction, clip + Adam one kernel,
    and the data-parallel exchange one RCCL all-reduce over the flat gradient buffer;
  * the whole step (about 600 kernel launches) is captured into a HIP graph per step kind (G / D);
    with world_size > 1 the graph is split around the (eager) all-reduce;
  * one process per GPU, pure data parallel over clips; BatchNorm statistics are per rank
    (bn_sync='local'); host-side random decisions come from identically seeded CPU generators so all
    ranks take the same D-vs-G branch (gan.py:105) and curriculum branch (JL:127).
"""
import os
import weakref

import torch
import torch.distributed as dist

from . import layers, ops, ops16

_ALIGN = 64  # elements: every parameter starts 256-B aligned inside the flat buffer


def _single_rank_dp():
  """Test aid: MS_DP_SINGLE_RANK=1 makes a ONE-rank process group take the data-parallel path (split graphs, eager RCCL
  all-reduce, broadcast), so that the RCCL calls of that path can be executed on a box with a single GPU."""
  return os.environ.get('MS_DP_SINGLE_RANK', '0') == '1'


def _dp_world(process_group=None):
  """Ranks of the data-parallel job; 1 = no exchange step."""
  if not (dist.is_available() and dist.is_initialized()):
    return 1
  world = dist.get_world_size(process_group)
  return 2 if (world == 1 and _single_rank_dp()) else world


def average_flat_gradients(flat_g, process_group=None, wire=None):
  """The data-parallel exchange step: one all-reduce (RCCL over xGMI on the GPUs; gloo in the CPU tests) of the flat
  gradient buffer, leaving the mean over ranks -- what a single device would have computed on the global batch of
  equally sized shards (up to per-rank BatchNorm statistics).
  wire: optional 16-bit buffer of the same length -- the gradients travel in its dtype (half the bytes on the links; every
  rank ends with the same values, so the replicas stay identical, but they are the rounded means)."""
  if not (dist.is_available() and dist.is_initialized()):
    return flat_g
  world = dist.get_world_size(process_group)
  if world == 1 and not _single_rank_dp():
    return flat_g
  buf = flat_g
  if wire is not None:
    wire.copy_(flat_g)
    buf = wire
  if dist.get_backend(process_group) == 'nccl':
    dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=process_group)
    if wire is not None:
      flat_g.copy_(wire)
  else:
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
    if wire is not None:
      flat_g.copy_(wire)
    flat_g.mul_(1.0 / world)
  return flat_g


def broadcast_from_rank0(tensors, process_group=None):
  """Start every rank from rank 0's values (parameters, BatchNorm buffers): with identical weights and identical
  averaged gradients the replicas stay bit-identical without any further parameter traffic."""
  if _dp_world(process_group) == 1:
    return
  for t in tensors:
    dist.broadcast(t, src=0, group=process_group)


def peek_step_decisions(D_prob, thresh_value, thresh_iters, thresh_num_iters, thresh_end):
  """What gan.py:105 (D-step vs G-step) and JL:127 (curriculum branch) will draw from the host generator, without
  consuming it.  Ranks seed their host generators identically, so every rank takes the same branch."""
  state = torch.get_rng_state()
  r_gan, r_branch = torch.rand(1).item(), torch.rand(1).item()
  torch.set_rng_state(state)
  kind = 'D' if r_gan < D_prob else 'G'
  thresh_now = (thresh_value if thresh_iters < thresh_num_iters else thresh_end) if kind == 'G' else thresh_value
  return kind, (kind == 'G' and r_branch > thresh_now)


class FlatAdam:
  """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8) + clip_grad_norm_(params, max_norm) over flat buffers."""

  def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0, order_last=(), order_first=()):
    self.params = [p for p in params if p.requires_grad]
    if not self.params:
      raise ValueError('no trainable parameters')
    # used-first layout: parameters named in `order_last` (sub-networks that are idle on the path being trained) sit at
    # the END of the flat buffers, so the live gradients form one prefix -- what the data-parallel exchange moves.
    # `order_first`: the parameters whose gradients are complete EARLIEST in the backward pass (the last layers of the forward
    # pass) lead the buffer: one contiguous bucket that can be exchanged while the rest of the backward pass runs.
    last = set(id(p) for p in order_last)
    first = set(id(p) for p in order_first) - last
    self.params = ([p for p in self.params if id(p) in first] + [p for p in self.params if id(p) not in last and id(p) not in first] +
                   [p for p in self.params if id(p) in last])
    self.n_first = sum(1 for p in self.params if id(p) in first)
    dev = self.params[0].device
    if dev.type != 'cuda':
      raise RuntimeError('FlatAdam runs on the MI355X only (parameters are on %s)' % dev)
    offs, total = [], 0
    for p in self.params:
      offs.append(total)
      total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    self.offsets, self.total = offs, total
    self.first_elems = offs[self.n_first] if self.n_first < len(offs) else total      # length of the `order_first` bucket
    self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
    self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
    self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
    self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
    self._grad_views = []
    with torch.no_grad():
      for p, o in zip(self.params, offs):
        if p.dtype != torch.float32:
          raise TypeError('FlatAdam: float32 parameters only')
        view = self.flat_p[o:o + p.numel()].view_as(p)
        view.copy_(p)
        p.data = view
        g = self.flat_g[o:o + p.numel()].view_as(p).data     # (.data: its own version counter, see gather_foreign_grads)
        p.grad = g
        p._ms_grad_slot = g            # kernels write the step's first gradient straight into the flat buffer
        p._ms_grad_fresh = False
        self._grad_views.append(g)
    self.lr, self.betas, self.eps, self.max_norm = lr, betas, eps, max_norm
    self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
    self.partials = torch.zeros(ops.lib().ms_reduce_partials_count(total), dtype=torch.float32, device=dev)
    self.step_state = torch.zeros(4, dtype=torch.int32, device=dev)
    # torch.optim.Adam keeps one step count per parameter, started at the parameter's first gradient (and skips
    # parameters that never had one): segment table for the flat buffer
    seg = torch.empty(total // _ALIGN, dtype=torch.int32)
    for i, (p, o) in enumerate(zip(self.params, offs)):
      seg[o // _ALIGN:(o + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN) // _ALIGN] = i
    self.seg_of_chunk = seg.to(dev)
    self.host_first = [-1] * len(self.params)          # global step (1-based) of each parameter's first gradient
    self.seg_first = torch.full((len(self.params),), -1, dtype=torch.int32, device=dev)
    self.seg_scratch = torch.zeros(2 * len(self.params), dtype=torch.float32, device=dev)
    self.host_step = 0                                 # optimizer steps executed so far
    self._grad_versions = [g._version for g in self._grad_views]

  def zero_grad(self):
    self.flat_g.zero_()
    for p in self.params:
      p._ms_grad_fresh = True
    self._grad_versions = [g._version for g in self._grad_views]

  def active_params(self):
    """Indices of the parameters that received a gradient since zero_grad(): their kernels wrote into the slot
    (`_ms_grad_fresh` cleared), or plain torch autograd accumulated into the slot view / replaced p.grad (e.g. the 'lin'
    style path: EmbLin's matmul, joint_late_cluster_soft_style.py:166) -- those are folded back first."""
    self.gather_foreign_grads()
    return [i for i, p in enumerate(self.params) if not p._ms_grad_fresh]

  def live_elems(self, active):
    """Length of the flat-buffer prefix that holds every gradient of the parameters `active` (used-first layout)."""
    end = 0
    for i in active:
      end = max(end, self.offsets[i] + (self.params[i].numel() + _ALIGN - 1) // _ALIGN * _ALIGN)
    return end

  def mark_active(self, indices):
    """Record first-gradient steps for parameters that get a gradient in the step about to run."""
    new = [i for i in indices if self.host_first[i] < 0]
    if new:
      for i in new:
        self.host_first[i] = self.host_step + 1
      self.seg_first.copy_(torch.tensor(self.host_first, dtype=torch.int32), non_blocking=False)

  def reset_state(self):
    self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.step_state.zero_()
    self.host_first = [-1] * len(self.params)
    self.seg_first.fill_(-1)
    self.host_step = 0

  def gather_foreign_grads(self):
    """Gradients that did not come through the kernels' write-through: p.grad replaced (model.zero_grad(set_to_none=True)
    then backward) is folded back; a slot view autograd ACCUMULATED into (version counter moved since zero_grad) is marked
    as having a gradient, so its Adam segment runs."""
    with torch.no_grad():
      for p, g, v0 in zip(self.params, self._grad_views, self._grad_versions):
        if p.grad is None:
          p.grad = g
        elif p.grad.data_ptr() != g.data_ptr():
          g.copy_(p.grad)
          p.grad = g
          p._ms_grad_fresh = False
        elif p._ms_grad_fresh and g._version != v0:
          p._ms_grad_fresh = False

  def clip_and_step(self, count=True):
    """total_norm = ||g||_2 over all parameters; g *= min(1, max_norm/(norm+1e-6)); Adam update (per-parameter step
    counts).  count=False while a HIP graph is being captured (nothing executes)."""
    ops.grad_norm(self.flat_g, self.norm, self.partials)
    ops.adam_step_segmented(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.norm, self.max_norm, self.lr,
                            self.betas[0], self.betas[1], self.eps, self.step_state, self.seg_of_chunk, self.seg_first,
                            self.seg_scratch)
    ops.refresh_prepared_weights(self.flat_p)        # data-gradient weights of all blocks of this network: one launch
    self.seen_version = self._param_versions()
    if count:
      self.host_step += 1

  def resync_if_modified(self):
    """Parameters changed behind the optimizer's back (load_state_dict, manual edits: torch bumps the version counter the
    views share with the flat buffer): rebuild what was derived from them before a captured step replays."""
    v = self._param_versions()
    if v != getattr(self, 'seen_version', None):
      ops.refresh_prepared_weights(self.flat_p)
      self.seen_version = v

  def _param_versions(self):
    # p.data was re-pointed into the flat buffer, so every parameter keeps its own version counter
    return sum(p._version for p in self.params) + self.flat_p._version

  @property
  def step_count(self):
    return int(self.step_state[0])


class MixStageTrainStep:
  """Runs reference-equivalent training steps for GAN(G, D) on one GPU or data-parallel over ranks."""

  def __init__(self, model, lr=1e-4, clip=1.0, use_graphs=True, process_group=None, time_steps=64, overlap_wgrad=False,
               bn_sync='local', overlap_allreduce=False, grad_buckets=None, grad_exchange='fp32'):
    self.model = model
    if bn_sync not in ('local', 'global'):
      raise ValueError("bn_sync must be 'local' or 'global'")
    self.bn_sync = bn_sync
    rccl = dist.is_available() and dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
    if bn_sync == 'global':
      if not rccl:
        use_graphs = False      # gloo's statistics exchanges (host side) cannot be captured; RCCL's are graph nodes like any kernel
    if getattr(model, '_ms_dt', 0) == ops16.MS_F16:
      # loss-mean gradients of ~1/(B*T*P) = 5e-6 sit in the fp16 subnormal range: without a loss scale (not implemented) the
      # activation gradients underflow.  fp16 is the inference arithmetic (BASELINE configs[4]); train in bf16 or fp32.
      raise NotImplementedError("training in fp16 needs loss scaling, which this path does not implement: use 'bf16' or 'fp32'")
    self.process_group = process_group
    if grad_exchange not in ('fp32', 'bf16'):
      raise ValueError("grad_exchange must be 'fp32' or 'bf16'")
    # 'bf16': the gradient all-reduce moves bf16 values (an option for the xGMI-bound exchange of the 16-bit mode; the optimizer
    # then sees the bf16-rounded mean gradient on every rank).  Default: fp32, the exchange whose result equals a single device's.
    self.grad_exchange = grad_exchange
    self._wire = {}
    ops.set_bn_sync(bn_sync == 'global', process_group)
    # Data-parallel exchange (world > 1): the live prefix of the flat gradient buffer in `grad_buckets` all-reduces, issued in
    # REVERSE order of the forward pass.  The first of them -- the decoder / logits / classifier gradients, which lead the
    # buffer (FlatAdam order_first) -- starts at the backward-pass marker behind the UNet, on a communication stream, next to
    # the rest of the backward pass; the others follow the end of the backward pass.  On RCCL the collectives are captured
    # into the step's HIP graph (one graph per step kind); gloo (CPU-side transport: the tests) keeps two graphs around an
    # eager exchange and no overlap.
    # overlap_allreduce is OFF by default: measured with ONE rank on the data-parallel form of the step (MS_DP_SINGLE_RANK=1, bf16
    # G-step, all of it RCCL's local reduce-copy of 60 MB): plain 2.69 ms, exchange captured on the step's own stream 2.79-2.81,
    # on the communication stream with the early bucket 2.97 -- the cross-stream edges inside the captured graph cost more than
    # a one-rank exchange can give back (the same was measured for weight gradients on a side stream).  Whether it pays with 8
    # ranks (an exchange of ~0.4 ms over xGMI) has to be measured on such a node; no N > 1 RCCL run exists so far.
    # MS_CAPTURE_ALLREDUCE=0: keep the gradient exchange outside the graphs (two graphs per step with an eager all-reduce between
    # them), should capturing the collective fail on some stack
    # DEFAULT: the exchange stays OUTSIDE the graphs (two graphs per step around an eager all-reduce -- the form every multi-rank
    # test has exercised).  MS_CAPTURE_ALLREDUCE=1 opts into the captured form; it has only ever run with one RCCL rank.
    self.capture_allreduce = rccl and os.environ.get('MS_CAPTURE_ALLREDUCE', '0') == '1'
    # (the overlapped exchange forks a communication stream off the backward pass: only inside the one captured graph; and its
    # early bucket must not race weight gradients still running on a side stream)
    self.overlap_allreduce = bool(overlap_allreduce) and rccl and self.capture_allreduce
    if self.overlap_allreduce and overlap_wgrad:
      raise ValueError('overlap_allreduce and overlap_wgrad are exclusive: the early bucket would be exchanged while its weight '
                       'gradients are still in flight on the side stream')
    # (buckets only pay when the exchange overlaps the backward pass; without the overlap ONE all-reduce of the live prefix has the
    # fewest launches and the longest transfers: one rank on RCCL, G-step +0.095 ms with 4 captured collectives)
    self.grad_buckets = max(1, int(grad_buckets)) if grad_buckets is not None else (4 if self.overlap_allreduce else 1)
    if self.overlap_allreduce and self.grad_buckets < 2:
      self.grad_buckets = 2       # the early bucket is a bucket of its own: with one bucket it would be exchanged twice
    # overlap_wgrad: weight gradients on a side HIP stream (ms_conv_block_bwd_overlap).  Measured on MI355X / ROCm 7.2
    # inside the captured step it is SLOWER (5.01 vs 4.67 ms/step: cross-stream edges in the HIP graph cost more than the
    # concurrency buys), so it is off by default.
    self.side_stream = torch.cuda.Stream() if overlap_wgrad else None
    ops.enable_prepared_weights(True)
    ops.enable_deferred_wgrad(True)
    # idle on the audio path: constructed for state_dict parity, trained only by other branches / never
    idle = [p for n, p in model.G.named_parameters()
            if n.split('.')[0] in ('text_encoder', 'pose_encoder', 'style_dec', 'style_dec_gr', 'concat_encoder', 'smoothen')]
    early = [p for n, p in model.G.named_parameters() if n.split('.')[0] in ('decoder', 'logits', 'classify_cluster')]
    self.optim_G = FlatAdam(model.G.parameters(), lr=lr, max_norm=clip, order_last=idle, order_first=early)
    self.optim_D = FlatAdam(model.D.parameters(), lr=lr, max_norm=clip)
    self.use_graphs = use_graphs
    self.time_steps = time_steps
    self.pg = process_group
    self.world = _dp_world(process_group)          # > 1: the data-parallel form of the step (see _single_rank_dp)
    if self.world > 1:
      broadcast_from_rank0([self.optim_G.flat_p, self.optim_D.flat_p] + [b for b in model.buffers()], process_group)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
      # ranks that share a GPU (test set-ups): a launch whose workgroups wait for each other needs the device to itself
      # (physical identity: launchers that set HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank make every rank see index 0)
      import socket
      props = torch.cuda.get_device_properties(torch.cuda.current_device())
      ident = getattr(props, 'uuid', None)
      ident = str(ident) if ident is not None else 'pci:%s:%s:%s' % (getattr(props, 'pci_domain_id', '?'), getattr(props, 'pci_bus_id', '?'),
                                                                      getattr(props, 'pci_device_id', torch.cuda.current_device()))
      mine = (socket.gethostname(), ident)
      seen = [None] * dist.get_world_size(process_group)
      dist.all_gather_object(seen, mine, group=process_group)
      if len(set(seen)) < len(seen):
        import warnings
        warnings.warn('mix_stage_amd: %d ranks share a GPU (%s): the in-launch forms (chained decoder, in-launch BatchNorm) need the device to '
                      'themselves and are switched OFF for this process -- expect lower throughput than one rank per GPU' % (len(seen) - len(set(seen)) + 1, ident))
        ops16.set_in_launch_meetings(False)
    self._capture_stream = torch.cuda.Stream()      # warm-up and capture of every step kind: its scratch / counters are the graphs'
    # Health, every step and without a device synchronisation: the optimizer REFUSES a step whose gradient norm is not finite
    # (ms_adam_step_segmented: weights and moments untouched; a block whose OWN batch statistics are not finite does not write them
    # into its running buffers -- blocks in front of the fault have moved theirs by then, as in any step) and counts it in its state words; a raised meeting error word is sticky on the device (every later launch on those
    # counters yields NaN, i.e. more refused steps).  The two counters travel to pinned host memory behind every step -- the same
    # D2H path `losses` take when the caller reads them -- and are looked at when their copy has completed (at most
    # `_HEALTH_LAG` steps later): on_bad_step = 'raise' (default) raises then, with the model state intact; 'skip' warns, re-arms
    # the meeting counters and goes on; 'degrade' does the same and, when the cause was a meeting that timed out (the launch did not
    # have the device to itself), switches the in-launch meetings off so that the following steps run on the per-block kernels.  health_every > 0 additionally forces the synchronising check_health() every so many steps.
    self.on_bad_step = 'raise'          # | 'skip' | 'degrade' (skip + switch the in-launch meetings off after a meeting timed out)
    self.degraded = False
    self.health_every = 0
    self._health_pin = torch.zeros(2, dtype=torch.int32).pin_memory()
    self._health_events = []
    self._health_seen = 0
    self.skipped_steps = 0
    self._steps = 0
    self._graphs = {}
    self._static = None
    self._mods = None
    self._comm = torch.cuda.Stream() if (self.world > 1 and self.overlap_allreduce) else None
    self._early_done = False            # this step's first bucket went out at the backward-pass marker
    weakref.finalize(self, ops.drop_trainer_caches, [self.optim_G.flat_p.untyped_storage().data_ptr(),
                                                     self.optim_D.flat_p.untyped_storage().data_ptr()])
    self.losses = None       # list of 0-dim device tensors of the last step (reference order)
    # d(sum of losses)/d(loss) = 1: one constant on the default stream, made before any capture or side-stream pass
    self._seed = torch.ones((), dtype=torch.float32, device=self.optim_G.flat_p.device)
    self.fake_pose = None

  # ---- the eager pieces ------------------------------------------------------------------------------------
  def _kwargs(self, style):
    return dict(input_modalities=self.model.input_modalities, desc='train', sample_flag=0, description='train',
                style=style, time_steps=self.time_steps)

  def _forward_backward(self, audio, labels, pose, style, kind=None):
    m = self.model
    # the reference zeroes every gradient (trainer.py:604); only the network that steps receives gradients in a step (the
    # other one is frozen or runs under no_grad), and its buffer was zeroed before its own last step: one memset, not two
    if kind is None:
      self.optim_G.zero_grad()
      self.optim_D.zero_grad()
    else:
      (self.optim_G if kind == 'G' else self.optim_D).zero_grad()
    fake, losses, _ = m([audio, labels], pose, **self._kwargs(style))
    dev_losses = [l for l in losses if l.is_cuda and l.requires_grad]
    ops.reset_deferred_wgrad()
    ops.set_backward_overlap(self.side_stream)      # weight gradients on a side stream, joined below
    try:
      torch.autograd.backward(dev_losses, [self._seed] * len(dev_losses))   # == sum(losses).backward()
    finally:
      ops.join_backward_overlap()
      ops.set_backward_overlap(None)
    return fake, losses

  def _with_marker(self, fn, *args):
    """fn(*args) with the backward-pass marker armed (data-parallel G-steps with the overlapped exchange)."""
    arm = self.world > 1 and self.overlap_allreduce
    if arm:
      self._early_done = False
      ops.set_backward_marker(self._backward_marker)
    try:
      return fn(*args)
    finally:
      if arm:
        ops.set_backward_marker(None)

  def _bucket_bounds(self, opt, n):
    """[lo, hi) element ranges of the live prefix [0, n), in the order they are exchanged: the `order_first` bucket, then the
    rest in `grad_buckets - 1` pieces from the END of the prefix (the audio encoder, first in the forward pass, goes last)."""
    if self.grad_buckets == 1:
      return [(0, n)] if n else []
    first = min(opt.first_elems, n) if opt.n_first else 0
    out = [(0, first)] if first else []
    k = max(1, self.grad_buckets - (1 if first else 0))
    step = ((n - first + k - 1) // k + _ALIGN - 1) // _ALIGN * _ALIGN if n > first else 0
    hi = n
    while hi > first:
      lo = max(first, hi - step)
      out.append((lo, hi))
      hi = lo
    return out

  def _backward_marker(self):
    """Runs inside the backward pass, behind the UNet (ops.backward_marker): G-steps of a data-parallel job start the exchange
    of the first bucket here.  The weight-gradient kernels queued so far (decoder, logits, classifier) are launched first."""
    opt = self.optim_G
    if self.world <= 1 or not self.overlap_allreduce or not self.model.G_flag or not opt.n_first or self._early_done:
      return
    ops._flush_deferred_wgrad()
    cur = torch.cuda.current_stream()
    self._comm.wait_stream(cur)
    with torch.cuda.stream(self._comm):
      average_flat_gradients(opt.flat_g[:opt.first_elems], self.pg,
                             self._wire_buffer(opt)[:opt.first_elems] if self.grad_exchange == 'bf16' else None)
    self._early_done = True

  def _all_reduce(self, opt, active=None):
    """The exchange step: mean over ranks of the live prefix of the flat gradient buffer (used-first layout: 59.7 MB of
    the generator's 80.6 MB on the audio branch; everything behind it is zero on every rank), bucket by bucket."""
    if self.world <= 1:
      return
    n = opt.live_elems(active) if active is not None else opt.total
    early, self._early_done = self._early_done and opt is self.optim_G, False
    cur = torch.cuda.current_stream()
    comm = self._comm if self.overlap_allreduce else None
    if comm is not None:
      comm.wait_stream(cur)
    for lo, hi in self._bucket_bounds(opt, n):
      if early and lo == 0 and hi == min(opt.first_elems, n):
        continue                          # already on its way since the backward-pass marker
      wire = self._wire_buffer(opt)[lo:hi] if self.grad_exchange == 'bf16' else None
      if comm is not None:
        with torch.cuda.stream(comm):
          average_flat_gradients(opt.flat_g[lo:hi], self.pg, wire)
      else:
        average_flat_gradients(opt.flat_g[lo:hi], self.pg, wire)
    if comm is not None:
      cur.wait_stream(comm)

  def _wire_buffer(self, opt):
    buf = self._wire.get(id(opt))
    if buf is None:
      buf = self._wire[id(opt)] = torch.empty(opt.total, dtype=torch.bfloat16, device=opt.flat_g.device)
    return buf

  def _peek_decisions(self):
    th = self.model.G.thresh
    return peek_step_decisions(self.model.D_prob, th.value, th.iters, th.num_iters, th.end)

  def _consume_decisions(self, kind):
    """What GAN.forward / G.forward do on the host per training step, for a step that is replayed from a graph: the two RNG
    draws, the curriculum clock and the lambda schedule (gan.py:103,105; JL:127)."""
    m = self.model
    m.lambda_D, m.lambda_gan = m.lambda_scheduler.step()
    torch.rand(1)
    torch.rand(1)
    m.G.thresh.step(kind == 'G')
    m.G_flag = kind == 'G'
    m.fake_flag = True

  def _ensure_train_mode(self):
    """model.train() (trainer.py:1104) without its cost: nn.Module.train() walks the module tree and re-assigns `training` through
    Module.__setattr__ on each of the ~320 modules -- 0.4 ms of the 0.53 ms a replayed step used to cost on the host
    (tools/prof_host.py).  The flags are read instead and train() runs only when one of them is off (an evaluation in between)."""
    mods = self._mods
    if mods is None:
      mods = self._mods = list(self.model.modules())
    for mod in mods:
      if not mod.training:
        self.model.train()
        self._mods = None           # (re-list next time: whoever switched modes may also have edited the tree)
        return

  # ---- public ----------------------------------------------------------------------------------------------
  def step(self, audio, labels, pose, style, kind=None, inputs_unchanged=False):
    """One training step.  kind=None follows the reference's coin flip (host generator); 'G'/'D' pins it.
    Returns the step kind.  self.losses / self.fake_pose hold device tensors (no host sync here).
    inputs_unchanged=True (graph mode only): the caller promises the four inputs hold the previous step's values, the
    copies into the captured step's static buffers are skipped."""
    m = self.model
    self._ensure_train_mode()
    # (module-level switch of the ops: another trainer built in this process may have set it differently)
    ops.set_bn_sync(self.bn_sync == 'global', self.process_group)
    if kind is not None:
      saved = m.D_prob
      m.D_prob = 1.1 if kind == 'D' else -1.0
    try:
      k, pose_branch = self._peek_decisions()
      self.optim_G.resync_if_modified()
      self.optim_D.resync_if_modified()
      if not self.use_graphs:
        m._lambda_host_writes = True        # (eager: forward() refreshes the device-side loss weights itself)
        self.fake_pose, self.losses = self._with_marker(self._forward_backward, audio, labels, pose, style, k)
        opt = self.optim_G if m.G_flag else self.optim_D
        active = opt.active_params()
        opt.mark_active(active)
        self._all_reduce(opt, active)
        opt.clip_and_step()
      else:
        self._graph_step(k, pose_branch, audio, labels, pose, style, inputs_unchanged)
    finally:
      if kind is not None:
        m.D_prob = saved
    self._steps += 1
    self._post_health()
    if self.health_every and self._steps % self.health_every == 0:
      self.check_health()
    return k

  _HEALTH_LAG = 4          # health copies the host may run ahead of before it waits for the oldest one
  _HEALTH_EVERY = 8        # steps between two copies of the health words

  def _post_health(self):
    """Behind every `_HEALTH_EVERY`-th step: both optimizers' refused-step counters -> pinned host memory (two 4-byte asynchronous
    copies on the step's stream, no synchronisation), then look at whatever has already arrived.  The counters are monotonic, so a
    refused step is seen at the next copy; every step would be 2 x (4.3 us + 4 us of launch gap) of the device's time per step
    (profiles/r05_timeline_fp32_gstep.json: the two __amd_rocclr_copyBuffer launches in front of every replay)."""
    if self._steps % self._HEALTH_EVERY:
      if self._health_events:
        self._poll_health(False)
      return
    self._health_pin[0:1].copy_(self.optim_G.step_state[3:4], non_blocking=True)
    self._health_pin[1:2].copy_(self.optim_D.step_state[3:4], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    self._health_events.append(ev)
    self._poll_health(len(self._health_events) > self._HEALTH_LAG)

  def _poll_health(self, wait_oldest=False):
    evs = self._health_events
    if wait_oldest and evs:
      evs[0].synchronize()
    arrived = False
    while evs and evs[0].query():
      evs.pop(0)
      arrived = True
    if arrived:
      count = int(self._health_pin[0]) + int(self._health_pin[1])      # (monotonic: a later step's copy only adds to it)
      if count > self._health_seen:
        self._bad_steps(count)
      elif count < self._health_seen:
        self._health_seen = count       # the device counters were zeroed (FlatAdam.reset_state): count from there

  def _bad_steps(self, count):
    new, self._health_seen = count - self._health_seen, count
    self.skipped_steps += new
    words = None
    if ops16.bn_sync_error():               # (synchronises; we are off the fast path here)
      words = ops16.bn_sync_words()
      ops16.bn_sync_clear()                 # re-arm: counters and the sticky error word back to zero
    why = ('an in-launch BatchNorm / decoder-chain meeting timed out (the launch did not have the GPU to itself; sync words %s)' % (words,)
           if words is not None else 'a non-finite gradient norm')
    msg = ('%d training step(s) were refused on the device: %s.  Weights and Adam moments were left untouched by those steps, and no '
           'non-finite batch statistic was written into a running BatchNorm buffer -- but blocks in FRONT of the fault (and the clean '
           'half of a refused discriminator step) did move their running statistics and batch counts as in any step.  The meeting '
           'counters are re-armed.  See MixStageTrainStep.check_health' % (new, why))
    if self.on_bad_step == 'raise':
      raise RuntimeError(msg)
    import warnings
    if self.on_bad_step == 'degrade' and words is not None and ops16.in_launch_meetings():
      # The launches whose workgroups meet inside the launch (chained decoder, clip-resident blocks, in-launch BatchNorm) need every
      # workgroup resident at once: 256 workgroups on 256 CUs for the headline decoder, zero margin.  Something else holds CUs on this
      # device (a co-tenant, a profiler's kernels, another stream): instead of refusing every step from now on, fall back to the
      # per-block forms -- same arithmetic to rounding, no meetings, slower -- and capture the steps again.
      ops16.set_in_launch_meetings(False)
      self._graphs = {}
      self.degraded = True
      msg += ('  on_bad_step="degrade": in-launch meetings are now OFF for this process (per-block kernels, lower throughput); the '
              'captured steps are re-captured on their next use.')
    warnings.warn(msg)

  def check_health(self):
    """Synchronising form of the per-step health check (call it before saving a checkpoint and at the end of an epoch): raises
    (on_bad_step='raise') if a step was refused since the last look -- a launch whose workgroups meet inside the launch (in-launch
    BatchNorm, chained decoder) gave up waiting and poisoned its outputs with NaN, or a gradient was not finite.  Causes of the
    former: another process or a large kernel on another stream held compute units the launch needed -- run one trainer per GPU, or
    switch the forms off (ops16.set_in_launch_meetings(False))."""
    torch.cuda.synchronize()
    self._health_events = []
    count = int(self.optim_G.step_state[3]) + int(self.optim_D.step_state[3])
    if count > self._health_seen:
      self._bad_steps(count)
    elif count < self._health_seen:
      self._health_seen = count         # (counters zeroed by FlatAdam.reset_state since the last look)
    ops16.check_meetings()                  # a meeting that expired outside a training step (e.g. an evaluation forward)

  def _graph_step(self, k, pose_branch, audio, labels, pose, style, inputs_unchanged=False):
    self.model._lambda_host_writes = False   # the captured loss kernels read the device tensor written below
    key = (k, pose_branch, tuple(audio.shape), tuple(pose.shape))
    if self._static is None or self._static['key_shapes'] != key[2:]:
      self._static = dict(key_shapes=key[2:], audio=audio.clone(), labels=labels.clone(), pose=pose.clone(),
                          style=style.clone())
      self._graphs = {}
    st = self._static
    pairs = []
    for name, src in (('audio', audio), ('labels', labels), ('pose', pose), ('style', style)):
      # ~2 MB of device-to-device copies per step.  Always done: torch's version counters do not see writes made through
      # .data, raw-pointer kernels or DLPack producers, so "same tensor object, same version" does not prove "same batch".
      # inputs_unchanged=True is the caller's explicit promise (e.g. a profiling loop over one fixed batch).
      if src.data_ptr() == st[name].data_ptr() or inputs_unchanged:
        continue
      if src.is_cuda and src.is_contiguous() and src.dtype == st[name].dtype and src.shape == st[name].shape:
        pairs.append((st[name], src))       # one launch for all of them
      else:
        st[name].copy_(src, non_blocking=True)
    ops.copy_multi(pairs)
    entry = self._graphs.get(key)
    opt = self.optim_G if k == 'G' else self.optim_D
    if entry is None:
      entry = self._capture(key, k, st, opt)       # capture executes nothing: replay below does the step
    else:
      self._consume_decisions(k)
    # this step's loss weights (the schedule moved on the host): into the device tensor the captured loss kernels read
    self.model.write_lambdas(opt.flat_p.device)
    for mod in entry['bn_tape']:
      mod.__dict__['_pending_batches'] += 1      # (a plain int attribute: nn.Module.__setattr__'s type checks cost 0.6 us each, x 53 blocks)
    opt.mark_active(entry['active'])
    entry['fwd_bwd'].replay()
    if entry['opt'] is not None:            # (gloo: the exchange runs eagerly between the two graphs)
      self._all_reduce(opt, entry['active'])
      entry['opt'].replay()
    opt.host_step += 1
    # prepared weights that did not exist when this graph was captured (blocks only the other step kind runs)
    ops.refresh_prepared_weights(opt.flat_p, start=entry['n_prepared'])
    self.fake_pose, self.losses = entry['fake'], entry['losses']

  def _capture(self, key, k, st, opt):
    m = self.model
    sched = m.lambda_scheduler
    if not (hasattr(sched, 'state') and hasattr(sched, 'set_state')):
      # the warm-up and capture passes below call scheduler.step(); their effect on the schedule has to be undone
      raise NotImplementedError('use_graphs=True needs a lambda scheduler with state() / set_state() (see '
                                'gan.IncrementalLambdaScheduler); or construct MixStageTrainStep(..., use_graphs=False)')
    sched_state = sched.state()
    m.write_lambdas(opt.flat_p.device)
    rng = torch.get_rng_state()
    thresh = (m.G.thresh.value, m.G.thresh.iters)
    bn_state = {n: b.clone() for n, b in m.named_buffers()}
    # one eager pass on a side stream sizes the workspace and warms the allocator, then undo its side effects
    side = self._capture_stream
    side.wait_stream(torch.cuda.current_stream())
    warm_tape = []
    layers.set_train_tape(warm_tape)
    try:
      with torch.cuda.stream(side):
        self._forward_backward(st['audio'], st['labels'], st['pose'], st['style'], k)
    finally:
      layers.set_train_tape(None)
    warm_active = opt.active_params()
    if self.world > 1 and self.capture_allreduce:
      # the collective library sets up channels / buffers the first time it sees a message size: let that happen in an eager
      # exchange of exactly the slices the captured step will exchange (the warm-up gradients are discarded anyway)
      with torch.cuda.stream(side):
        self._early_done = False
        self._all_reduce(opt, warm_active)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.no_grad():
      for n, b in m.named_buffers():
        b.copy_(bn_state[n])
    for mod in warm_tape:
      mod._pending_batches -= 1
    torch.set_rng_state(rng)
    m.G.thresh.value, m.G.thresh.iters = thresh
    sched.set_state(sched_state)
    tape = []
    layers.set_train_tape(tape)
    # with a process group alive, RCCL's watchdog thread issues HIP calls of its own: only this thread's calls may
    # invalidate the capture
    mode = 'thread_local' if self.world > 1 else 'global'
    one_graph = self.world == 1 or self.capture_allreduce
    g1 = torch.cuda.CUDAGraph()
    try:
      # (a capture that fails half-way cannot be retried in this process -- the stream stays in capture mode -- so there is no
      # automatic fallback here: MS_CAPTURE_ALLREDUCE=0 selects the two-graph form with the eager exchange up front)
      with torch.cuda.graph(g1, stream=side, capture_error_mode=mode):
        if one_graph and self.world > 1:
          fake, losses = self._with_marker(self._forward_backward, st['audio'], st['labels'], st['pose'], st['style'], k)
          # (the warm-up pass above ran the same forward / backward: its set of parameters with gradients is this step's)
          self._all_reduce(opt, warm_active)
        else:
          fake, losses = self._forward_backward(st['audio'], st['labels'], st['pose'], st['style'], k)
        if one_graph:
          opt.clip_and_step(count=False)
    finally:
      layers.set_train_tape(None)
    active = opt.active_params()
    for mod in tape:                      # the capture pass itself ran no kernels
      mod._pending_batches -= 1
    g2 = None
    if not one_graph:
      g2 = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g2, stream=side, capture_error_mode=mode):
        opt.clip_and_step(count=False)
    entry = dict(fwd_bwd=g1, opt=g2, fake=fake, losses=losses, bn_tape=tape, active=active,
                 n_prepared=ops.prepared_count(opt.flat_p))
    self._graphs[key] = entry
    return entry

  def state_checksums(self):
    """(sum, l2) of the flat parameter buffers -- cheap parity probe."""
    return {n: (float(o.flat_p.double().sum()), float(o.flat_p.double().norm()))
            for n, o in (('G', self.optim_G), ('D', self.optim_D))}
