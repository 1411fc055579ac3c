"""GPU: train-mode BatchNorm INSIDE the conv launch (16-bit path, EP_BN_FUSED; layers.py:77-78 as one HBM pass).

The workgroups that share a channel tile exchange their partial batch statistics inside the launch.  Checked here: against the
two-launch form on the same inputs (statistics to fp32 rounding, y_raw bit for bit, y to one 16-bit rounding), bitwise
repeatability under uneven load on the chip (a stale or torn partial would change the statistics), the counters left zeroed,
no time-out raised."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
BN_TRAIN, PLAIN, BCAST, UP2 = 2, 0, 1, 2


def _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, fused, seed=0, dt=torch.bfloat16, reps=1, tensors=None, sync=True,
               big_bias=None):
  from mix_stage_amd import ops, ops16
  from mix_stage_amd._lib import MS_BF16, MS_F16, FwdOptions, check, lib
  L = lib()
  msdt = MS_BF16 if dt == torch.bfloat16 else MS_F16
  prev = L.ms_debug_set_bn_fused(1 if fused else 0)
  try:
    if tensors is None:
      g = torch.Generator().manual_seed(seed)
      sp = (H, W) if nd == 2 else (W,)
      cin_tot = cin if in_mode == BCAST else cin * groups
      kt = (k, k) if (nd == 2 and not isinstance(k, tuple)) else (k if isinstance(k, tuple) else (k,))
      fan = cin
      for v in kt:
        fan *= v
      ctot = cout * groups
      w = (torch.randn((ctot, cin) + tuple(kt), generator=g) * fan ** -0.5).to(DEV)
      bias = (torch.randn(ctot, generator=g) * 0.1).to(DEV)
      if big_bias is not None:
        bias[big_bias[0]] = big_bias[1]
      gamma = (0.5 + torch.rand(ctot, generator=g)).to(DEV)
      beta = (torch.randn(ctot, generator=g) * 0.1).to(DEV)
      rm = (torch.randn(ctot, generator=g) * 0.1).to(DEV)
      rv = (0.5 + torch.rand(ctot, generator=g)).to(DEV)
      if in_mode == UP2:
        x = ops16.to_cb8(torch.randn((B, cin_tot, W // 2), generator=g).to(DEV) + 0.3, msdt)
        x2 = ops16.to_cb8(torch.randn((B, cin_tot, W), generator=g).to(DEV), msdt)
      else:
        x = ops16.to_cb8(torch.randn((B, cin_tot) + sp, generator=g).to(DEV) + 0.3, msdt)
        x2 = None
      tensors = (w, bias, gamma, beta, rm, rv, x, x2)
    w, bias, gamma, beta, rm, rv, x, x2 = tensors
    rm, rv = rm.clone(), rv.clone()
    ctot = w.shape[0]
    geom = ops.ConvGeom(nd, groups, k, s, p)
    d = geom.desc(B, cin, H, W, cout, BN_TRAIN, in_mode, msdt)
    sp_out = (d.OH, d.OW) if nd == 2 else (d.OW,)
    c8 = (ctot + 7) // 8
    syncbuf = ops16._ensure_bn_sync(x.device)
    ws = ops.workspace(d._fwd_ws, x.device)
    outs = []
    P = ops._ptr
    for _ in range(reps):
      y = torch.full((B, c8) + sp_out + (8,), float('nan'), dtype=dt, device=DEV)
      y_raw = torch.full_like(y, float('nan'))
      save = torch.full((4 * ctot,), float('nan'), dtype=torch.float32, device=DEV)
      opt = FwdOptions(None, syncbuf.data_ptr(), syncbuf.numel())
      check(L.ms_conv_block_fwd_ex(ctypes.byref(d), P(x), P(x2), P(w), P(bias), P(gamma), P(beta), P(rm), P(rv), P(y_raw), P(y),
                                   P(save), P(ws), ws.numel(), ops._stream(), ctypes.byref(opt)), 'ms_conv_block_fwd_ex')
      outs.append((y, y_raw, save))
    if sync:
      torch.cuda.synchronize()
    return outs, (rm, rv), tensors
  finally:
    L.ms_debug_set_bn_fused(prev)


def _labels_of(fn):
  from mix_stage_amd import ops
  ops.timing_enable(True)
  try:
    fn()
    torch.cuda.synchronize()
    return [r['label'] for r in ops.timing_report()]
  finally:
    ops.timing_enable(False)


GEOMS = [
    # name, nd, B, cin, cout, groups, k, s, p, H, W, in_mode
    ('decoder_headline', 1, 32, 256, 256, 8, 3, 1, 1, 1, 64, PLAIN),     # 512 workgroups of 64 x 128, 16 per group, 2 per CU
    ('decoder0_bcast', 1, 32, 266, 256, 8, 3, 1, 1, 1, 64, BCAST),
    ('unet_t64', 1, 32, 256, 256, 1, 3, 1, 1, 1, 64, PLAIN),             # 64 x 64 tiles, 32 per group
    ('unet_down_t32', 1, 32, 256, 256, 1, 4, 2, 1, 1, 64, PLAIN),
    ('unet_up2', 1, 32, 256, 256, 1, 3, 1, 1, 1, 32, UP2),               # register-staged kernel (no LDS-DMA)
    ('ragged', 1, 5, 72, 40, 1, 3, 1, 1, 1, 50, PLAIN),                  # partial tiles, channels not a multiple of 8
    ('deep_t2', 1, 6, 64, 64, 1, 4, 2, 1, 1, 4, PLAIN),                  # a single tile per channel tile
    ('ae_deep_2d', 2, 8, 128, 256, 1, 3, 1, 1, 8, 15, PLAIN),
    ('m4_decoder', 1, 32, 256, 256, 4, 3, 1, 1, 1, 64, PLAIN),
]


@pytest.mark.parametrize('geo', GEOMS, ids=[g[0] for g in GEOMS])
def test_fused_matches_two_launch_form(geo):
  _, nd, B, cin, cout, groups, k, s, p, H, W, in_mode = geo
  labels = _labels_of(lambda: _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True))
  assert any('+bnfused' in l for l in labels), labels
  assert not any('bn_finalize' in l or 'bn_apply' in l for l in labels), labels
  (f,), (rm_f, rv_f), tensors = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True)
  (u,), (rm_u, rv_u), _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, False, tensors=tensors)
  y_f, raw_f, save_f = f
  y_u, raw_u, save_u = u
  ctot = cout * groups
  assert not torch.isnan(save_f).any() and not torch.isnan(y_f.float()).any()
  # y_raw: the in-launch form keeps it only for channel blocks whose BatchNorm + LeakyReLU map does not invert safely (none with
  # these parameters: the buffer stays as it was); where written it is the same accumulators with the same rounding
  written = ~torch.isnan(raw_f.float())
  assert not written.any()
  # statistics: the same per-tile partials merged in fp64 (Chan) on both sides -> fp32 rounding at most
  mean_f, mean_u = save_f[:ctot], save_u[:ctot]
  inv_f, inv_u = save_f[ctot:2 * ctot], save_u[ctot:2 * ctot]
  assert (mean_f - mean_u).abs().max().item() <= 2e-6 * (1 + mean_u.abs().max().item())
  assert ((inv_f - inv_u).abs() / inv_u.abs()).max().item() <= 2e-6
  assert (save_f[2 * ctot:] - save_u[2 * ctot:]).abs().max().item() <= 1e-5 * (1 + save_u[2 * ctot:].abs().max().item())
  assert (rm_f - rm_u).abs().max().item() <= 1e-6 and ((rv_f - rv_u).abs() / rv_u.abs()).max().item() <= 2e-6
  # y: the fused form normalises the fp32 accumulators, the two-launch form the 16-bit y_raw: within one rounding of y_raw
  # through the affine map (|scale| * |y_raw| * 2^-8) plus one rounding of y
  sc = save_u[2 * ctot:3 * ctot].abs().max().item()
  bound = (raw_u.float().abs().max().item() * sc + y_u.float().abs().max().item()) * 2 ** -8 + 1e-6
  assert (y_f.float() - y_u.float()).abs().max().item() <= bound
  # and against the definition, from the two-launch form's y_raw and the statistics, elementwise to the same bound
  from mix_stage_amd import ops16
  raw32 = ops16.from_cb8(raw_u, ctot)
  shape = (1, -1) + (1,) * (raw32.dim() - 2)
  z = raw32 * save_f[2 * ctot:3 * ctot].view(shape) + save_f[3 * ctot:].view(shape)
  ref = torch.where(z > 0, z, 0.2 * z)
  assert (ops16.from_cb8(y_f, ctot) - ref).abs().max().item() <= bound
  if ctot % 8:
    assert float(y_f[:, -1, ..., ctot % 8:].float().abs().max()) == 0.0          # pad channels stay zero


def test_fused_is_bitwise_repeatable_under_uneven_load():
  """The hand-off (sc1 partial stores -> counter -> sc1 loads) must deliver every workgroup the SAME, COMPLETE set of partials.
  Three different input sets are launched in rotation, 30 times, next to a second stream that hammers HBM and the L2s; every
  output is compared bit for bit with the first launch of its input set, and every workgroup's statistics are recovered from
  y and compared with the block's.  A stale line (the previous launch's partials live at the same addresses and differ), a
  torn or an early read would move the statistics of some channel tile."""
  from mix_stage_amd import ops16
  geo = GEOMS[0][1:]
  nd, B, cin, cout, groups, k, s, p, H, W, in_mode = geo
  sets = []
  for seed in (11, 12, 13):
    (first,), _, tensors = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, seed=seed)
    sets.append((first, tensors))
  assert not torch.equal(sets[0][0][2], sets[1][0][2])
  side = torch.cuda.Stream()
  big = torch.randn(64 << 20, device=DEV)
  with torch.cuda.stream(side):
    for i in range(16):
      big = big * 1.0001 + 0.5            # 512 MB of traffic per pass beside the conv launches
  for rep in range(30):
    first, tensors = sets[rep % 3]
    (out,), _, _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, tensors=tensors)
    y, raw, save = out
    assert torch.equal(save, first[2]), rep
    assert torch.equal(y.view(torch.int16), first[0].view(torch.int16)), rep
  torch.cuda.synchronize()
  # every tile of y was normalised with the block's statistics (each workgroup derives them itself from the partials it read)
  ctot = cout * groups
  for first, tensors in sets:
    (unf,), _, _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, False, tensors=tensors)
    y32, raw32 = ops16.from_cb8(first[0], ctot), ops16.from_cb8(unf[1], ctot)
    z = raw32 * first[2][2 * ctot:3 * ctot].view(1, -1, 1) + first[2][3 * ctot:].view(1, -1, 1)
    ref = torch.where(z > 0, z, 0.2 * z)
    err = (y32 - ref).abs()
    assert err.max().item() <= (ref.abs().max().item() + raw32.abs().max().item()) * 2 ** -8
    assert (err / (ref.abs() + 1e-2)).mean().item() <= 2 ** -8            # no tile with shifted statistics
  assert not ops16.bn_sync_error()
  # arrive / depart counters re-armed (the per-block buffers of the fp32 clip kernels -- keys tagged 'block', other test files -- count
  # monotonically and are not part of this protocol)
  assert all(int(b.abs().sum().item()) == 0 for k, b in ops16._bn_sync.items() if 'block' not in k and 'chain' not in k)


def test_large_grids_keep_the_two_launch_form():
  """More workgroups than the device holds at once (first audio-encoder layer): conv + statistics, then the normalising launch."""
  labels = _labels_of(lambda: _run_block(2, 32, 1, 64, 1, 3, 1, 1, 64, 128, PLAIN, True))
  assert not any('+bnfused' in l for l in labels), labels
  assert any('+bnstats' in l for l in labels), labels


def test_unsafe_channel_blocks_keep_y_raw():
  """gamma ~ 0 or |beta| >> |gamma|: x_hat cannot be recovered from y; exactly those 8-channel blocks get their y_raw written by the
  in-launch form (bit-identical to the two-launch form's), the others do not."""
  nd, B, cin, cout, groups, k, s, p, H, W, in_mode = GEOMS[2][1:]
  (_,), _, tensors = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True)
  w, bias, gamma, beta, rm, rv, x, x2 = tensors
  gamma = gamma.clone(); beta = beta.clone()
  gamma[3] = 1e-5                      # block 0
  beta[42] = 5.0; gamma[42] = 0.3      # block 5
  tensors = (w, bias, gamma, beta, rm, rv, x, x2)
  (f,), _, _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, tensors=tensors)
  (u,), _, _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, False, tensors=tensors)
  written = ~torch.isnan(f[1].float())                       # (B, C8, T, 8)
  per_block = written.flatten(2).all(-1).all(0)
  some = written.flatten(2).any(-1).any(0)
  assert torch.equal(per_block, some)                        # a block is written whole or not at all
  assert per_block.nonzero().flatten().tolist() == [0, 5]
  for cb in (0, 5):
    assert torch.equal(f[1][:, cb].view(torch.int16), u[1][:, cb].view(torch.int16))


def test_channel_with_mean_far_above_sigma_keeps_its_variance():
  """A channel whose mean is 1000 sigma because of its bias: the tile partials are sums of the bias-free accumulators, published
  as (mean = bias + sum / n, M2, count) and combined in fp64 -- fp32 sum x^2 - (sum x)^2 / n over x = acc + bias would lose the
  variance entirely.  Reference: the same 16-bit operands in float64."""
  from mix_stage_amd import ops16
  nd, B, cin, cout, groups, k, s, p, H, W, in_mode = GEOMS[0][1:]
  ch = 517
  (f,), (rm, rv), tensors = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, seed=5, big_bias=(ch, 1000.0))
  w, bias, gamma, beta, rm0, rv0, x, x2 = tensors
  ctot = cout * groups
  x64 = ops16.from_cb8(x, cin * groups).double()
  w64 = w.to(torch.bfloat16).double()
  ref = torch.nn.functional.conv1d(x64, w64, bias.double(), stride=s, padding=p, groups=groups)
  mean = ref.mean((0, 2)); var = ref.var((0, 2), unbiased=False)
  save = f[2].double()
  assert abs(float(mean[ch]) - 1000.0) < 5 and float(var[ch].sqrt()) < 2.0              # mean / sigma ~ 1e3
  assert (save[:ctot] - mean).abs().max().item() <= 1e-4 * (1 + mean.abs().max().item())
  inv_ref = 1.0 / torch.sqrt(var + 1e-5)
  rel = ((save[ctot:2 * ctot] - inv_ref).abs() / inv_ref)
  assert rel.max().item() <= 2e-3, (rel.max().item(), int(rel.argmax()))
  assert rel[ch].item() <= 2e-3
  # the normalised output of that channel has unit variance (it would be garbage with a lost variance)
  y = ops16.from_cb8(f[0], ctot)[:, ch].double()
  z = torch.where(y > 0, y, y / 0.2)
  xhat = (z - float(beta[ch])) / float(gamma[ch])
  assert abs(float(xhat.var(unbiased=False)) - 1.0) < 0.1
  assert not ops16.bn_sync_error()


def test_two_streams_run_fused_blocks_concurrently():
  """Two HIP streams launching in-launch-BatchNorm blocks at the same time: each stream has its own counters and scratch
  (ms_fwd_options.bn_sync, ops.workspace per stream), so neither sees the other's arrivals or partials.  Every output must equal
  the block's single-stream result bit for bit, and no meeting may time out."""
  from mix_stage_amd import ops16
  geo_a, geo_b = GEOMS[2][1:], GEOMS[8][1:]            # 128 + 256 workgroups: both launches fit on the chip together
  firsts = []
  for geo, seed in ((geo_a, 21), (geo_b, 22)):
    nd, B, cin, cout, groups, k, s, p, H, W, in_mode = geo
    (first,), _, tensors = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, seed=seed)
    firsts.append((geo, first, tensors))
  sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
  torch.cuda.synchronize()
  outs = {0: [], 1: []}
  for rep in range(20):
    for i, st in enumerate((sa, sb)):
      geo, first, tensors = firsts[i]
      nd, B, cin, cout, groups, k, s, p, H, W, in_mode = geo
      with torch.cuda.stream(st):
        (o,), _, _ = _run_block(nd, B, cin, cout, groups, k, s, p, H, W, in_mode, True, tensors=tensors, sync=False)
      outs[i].append(o)
  torch.cuda.synchronize()
  assert len(ops16._bn_sync) >= 3                      # the default stream's buffer and one per side stream
  for i in (0, 1):
    first = firsts[i][1]
    for o in outs[i]:
      assert torch.equal(o[2], first[2])
      assert torch.equal(o[0].view(torch.int16), first[0].view(torch.int16))
  assert not ops16.bn_sync_error()
