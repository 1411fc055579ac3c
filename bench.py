#!/usr/bin/env python3
"""Headline benchmark: Mix-StAGE train-step clips/sec (B=32 per GPU, T=64, 128-mel, 104-dim pose, M=S=8).

    python bench.py --gpus N --steps K --warmup W [--precision fp32|bf16x6|bf16]

With N > 1 and no launcher in the environment the script starts the N ranks itself (fresh child processes of
`python -m torch.distributed.run`, started BEFORE this process touches the GPU); under a launcher (RANK / WORLD_SIZE set) it
is one rank.  One rank per GPU, RCCL (`nccl`) over xGMI.

One "step" = one reference train step (trainer.py:604-674 contract, see mix_stage_amd/train_step.py): zero_grad ->
GAN.forward (D-step or G-step by the reference's seeded host coin flip, gan.py:105) -> backward -> gradient all-reduce
(N>1) -> clip_grad_norm_(.,1) -> Adam(1e-4).  Inputs are synthetic, resident in HBM before the timed region; weights are
the name-keyed deterministic fill.  The curriculum is pinned to the audio branch (thresh = 1), as stated in
BASELINE.md.  Prints ONE JSON line on rank 0: `value` over exactly K timed steps of the coin-flip sequence, plus (outside
the timed region) G-step and D-step times measured separately and their 50/50 blend.
"""
import argparse
import hashlib
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

B_PER_GPU, T, F_MEL, P, M, S = 32, 64, 128, 104, 8, 8
# --config: the default is the metric's own configuration (BASELINE.json `metric`); the others are BASELINE.json configs[1], [3]
# and [4] at their stated size and dtype on ONE GPU (profiles/r03_bench_c*.json; the driver runs the default)
CONFIGS = {
    'headline': dict(B=32, T=64, M=8, S=8, precision=None, kind='train',
                     metric='train-step clips/sec (B=32, T=64, M=8)'),
    'c2': dict(B=32, T=64, M=4, S=4, precision='bf16', kind='train',
               metric='train-step clips/sec (configs[1]: M=4, B=32, T=64, bf16, one MI355X)'),
    'c4': dict(B=32, T=256, M=25, S=25, precision='bf16', kind='train',
               metric='train-step clips/sec (configs[3]: M=25, T=256, bf16; one rank\'s shard B=32 on one MI355X)'),
    'c5': dict(B=1024, T=64, M=8, S=8, precision='fp16', kind='infer',
               metric='inference clips/sec (configs[4]: style transfer, B=1024, M=8, fp16, BN folded, HIP-graph replay)'),
}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16
# forward GFLOP per 32 clips: a G-step is ~3x the generator+D forward, a D-step ~1x G forward + 3x (2 D passes)
G_STEP_GFLOP, D_STEP_GFLOP = 3 * 99.0, 99.0 + 3 * 2 * 0.215


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=60)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--no-graphs', action='store_true', help='eager launches instead of HIP-graph replay')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-kernel-timing', action='store_true')
  ap.add_argument('--no-per-kind', action='store_true', help='skip the separate G-step / D-step timing')
  ap.add_argument('--no-bf16-extra', action='store_true', help='skip the extra bf16-mode measurement attached to the fp32 line')
  ap.add_argument('--seed', type=int, default=4321)
  ap.add_argument('--config', default='headline', choices=sorted(CONFIGS), help='headline (default, the metric) | c2 | c4 | c5: BASELINE configs[1], [3], [4]')
  ap.add_argument('--precision', default=None, choices=['fp32', 'bf16x6', 'bf16'],
                  help='fp32: exact fp32 matrix products (default, the parity headline); bf16x6: both operands split exactly '
                       'into 3 bf16 parts, 6 of 9 partial products on the bf16 pipe, fp32 accumulate (same measured accuracy); '
                       'bf16: native bf16 operands and bf16 activations in HBM, fp32 accumulate and BN statistics '
                       '(BASELINE configs[1]/[3] arithmetic; pose L1 is reported, not gated at 1e-4)')
  ap.add_argument('--bn-sync', default='local', choices=['local', 'global'])
  ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI); gloo only for smoke-testing the DP path')
  ap.add_argument('--same-device', action='store_true', help='smoke test: all ranks share cuda:0 (needs --dist-backend gloo)')
  ap.add_argument('--worker', action='store_true', help='(internal) this process is a rank worker started by the supervising rank process')
  ap.add_argument('--batch', type=int, default=None, help='diagnostic: clips per GPU instead of the configuration\'s (profiles/r04_frac_vs_batch.json); the default invocation is the metric\'s B=32')
  args = ap.parse_args()
  cfg = CONFIGS[args.config]
  global B_PER_GPU, T, M, S, G_STEP_GFLOP, D_STEP_GFLOP
  B_PER_GPU, T, M, S = cfg['B'], cfg['T'], cfg['M'], cfg['S']
  if args.batch:
    B_PER_GPU = args.batch
  if args.precision is None:
    args.precision = cfg['precision'] or 'fp32'
  if args.config != 'headline' or args.batch:
    # forward GFLOP of this configuration (SURVEY.md A.3 scaling: everything is linear in B*T, decoder and logits in M too)
    bt = (B_PER_GPU * T) / (32.0 * 64.0)
    fwd = bt * (62.31 + 4.21 + 0.50 + 4.87 + (26.02 + 0.87) * M / 8.0 + 0.215)
    G_STEP_GFLOP, D_STEP_GFLOP = 3 * fwd, fwd + 3 * 2 * 0.215 * bt
    args.no_bf16_extra = True
    args.no_cpu_baseline = True     # (the CPU baseline belongs to the metric's configuration)
  return args


def self_launch(args):
  """--gpus N without a launcher: start the N ranks as fresh children.  Nothing in this process has touched the GPU yet
  (argparse only), so no initialised HIP state is inherited or exec'd over."""
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.run(cmd, env=env).returncode


class Watchdog:
  """Rank worker: exits non-zero (code 3) when no progress was reported for `limit` seconds -- a hung collective must end the run
  with an error line, not hang the node.  beat(phase, limit) re-arms it."""

  def __init__(self):
    import threading
    self.last, self.limit, self.phase = time.time(), 900.0, 'start'
    t = threading.Thread(target=self._run, daemon=True)
    t.start()

  def beat(self, phase=None, limit=None):
    self.last = time.time()
    if phase:
      self.phase = phase
    if limit:
      self.limit = float(limit)

  def _run(self):
    while True:
      time.sleep(2.0)
      if time.time() - self.last > self.limit:
        sys.stderr.write('bench watchdog: no progress for %.0f s in phase %r -- exiting\n' % (self.limit, self.phase))
        sys.stderr.flush()
        os._exit(3)


def supervise(args):
  """One rank under the launcher (world > 1), BEFORE anything touches the GPU: runs the real rank as a fresh child process, so that a
  failed or hung first attempt -- this is the first code path of the repository to move bytes between GPUs -- can be repeated once
  in a more conservative form instead of losing the measurement.  The supervisors of all ranks agree over a gloo group (CPU) on
  whether the attempt succeeded.  Attempt 1: the default step (HIP graphs around an eager RCCL all-reduce).  Attempt 2 (only if any
  rank failed): no HIP graphs at all (`"dp_fallback": "no-graphs"` in the line).  The supervisor never initialises HIP."""
  import tempfile
  import torch.distributed as dist
  import datetime
  rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
  dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=1800))
  attempts = [({}, None), ({'MS_BENCH_NO_GRAPHS': '1'}, 'no-graphs')]
  tails = []
  for extra, tag in attempts:
    port = [0]
    if rank == 0:
      with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    # (a rendezvous of their own: under torchrun the agent hosts the store of the ORIGINAL port; the children's rank 0 hosts theirs)
    env = dict(os.environ, MASTER_PORT=str(port[0]), TORCHELASTIC_USE_AGENT_STORE='False', **extra)
    if tag:
      env['MS_DP_FALLBACK'] = tag
    errf = tempfile.TemporaryFile(mode='w+')
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--worker'] + ['--worker'],
                             env=env, stdout=subprocess.PIPE, stderr=errf, text=True)
    try:
      out, _ = child.communicate(timeout=1500)
      ok = child.returncode == 0
    except subprocess.TimeoutExpired:
      child.kill()
      out, _ = child.communicate()
      ok = False
    errf.seek(0)
    err_tail = errf.read()[-1500:]
    flags = [None] * world
    dist.all_gather_object(flags, (bool(ok), child.returncode, err_tail if not ok else ''))
    if all(f[0] for f in flags):
      if rank == 0:
        sys.stdout.write(out)
        sys.stdout.flush()
      dist.barrier()
      dist.destroy_process_group()
      return 0
    tails.append({'attempt': tag or 'default', 'ranks': [{'rank': i, 'rc': f[1], 'stderr_tail': f[2]} for i, f in enumerate(flags) if not f[0]]})
    sys.stderr.write('bench supervisor rank %d: attempt %r failed on %d rank(s)\n' % (rank, tag or 'default', sum(1 for f in flags if not f[0])))
  if rank == 0:
    print(json.dumps({'metric': CONFIGS[args.config]['metric'], 'value': None, 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps,
                      'warmup': args.warmup, 'error': 'every attempt failed', 'attempts': tails}))
  dist.destroy_process_group()
  return 1


def build_model(dev, precision='fp32'):
  import torch
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O     # deterministic weight fill + synthetic inputs (shared with the tests)
  G = A.JointLateClusterSoftStyle4_G(time_steps=T, out_feats=P, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  D = A.Speech2Gesture_D(in_channels=P)
  model = A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)
  model.load_state_dict(O.deterministic_state(model.state_dict()))
  model.G.thresh.value, model.G.thresh.iters = 1, 10 ** 9
  model = model.to(dev)
  if precision == 'bf16':
    A.set_compute_dtype(model, 'bf16')
  return model


def usable_cores():
  """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
    if quota != 'max':
      n = min(n, max(1, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return max(1, n)


def cpu_baseline(seed):
  """The oracle (pure PyTorch on the host cores) on a bounded sample of the same workload: fp32 G-steps and D-steps at
  B=32 after one untimed step of each kind (5 timed steps per kind, the MEDIAN is reported; 1 if a step takes longer than 4 s);
  plus 3 steps of each kind in fp64, the reference's own training dtype (trainer.py:138).  ~12 s of CPU work on 16 cores."""
  import torch
  from oracle import mixstage_oracle as O
  cores = min(usable_cores(), 32)       # oversubscribed intra-op threads make PyTorch CPU convs much slower
  torch.set_num_threads(cores)
  res = {}
  for name, dtype, n_max in (('fp32', torch.float32, 5), ('fp64', torch.float64, 3)):
    model = O.build_gan(M=M, S=S, T=T, P=P, dtype=dtype)
    og = torch.optim.Adam(model.G.parameters(), lr=1e-4)
    od = torch.optim.Adam(model.D.parameters(), lr=1e-4)
    audio, pose, labels, style = O.synthetic_batch(B_PER_GPU, T=T, F_=F_MEL, P=P, M=M, S=S, seed=1234, dtype=dtype)
    times, reps = {}, {}
    for kind in ('G', 'D'):
      t0 = time.perf_counter()
      O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
      warm = time.perf_counter() - t0
      n = n_max if warm < 4.0 else 1
      each = []
      for _ in range(n):
        t0 = time.perf_counter()
        O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
        each.append(time.perf_counter() - t0)
      times[kind], reps[kind] = sorted(each)[len(each) // 2], n         # median of the timed steps
    res[name] = (times, reps)
  (t32, r32), (t64, r64) = res['fp32'], res['fp64']
  blend32 = 0.5 * (t32['G'] + t32['D'])       # D_prob = 0.5 (gan.py:27)
  blend64 = 0.5 * (t64['G'] + t64['D'])
  return dict(value=round(B_PER_GPU / blend32, 2), unit='clips/s', cores=cores, kind='port',
              sample='oracle (PyTorch CPU fp32, %d threads): median of %d G-steps + %d D-steps at B=32 after 1 warm-up each; '
                     'G %.3f s, D %.3f s per step, 50/50 blend' % (cores, r32['G'], r32['D'], t32['G'], t32['D']),
              fp64=dict(value=round(B_PER_GPU / blend64, 2), unit='clips/s',
                        sample='same oracle in float64 (the reference trains in fp64): median of %d G + %d D step(s) after 1 warm-up each; '
                               'G %.3f s, D %.3f s' % (r64['G'], r64['D'], t64['G'], t64['D'])))


def measured_peaks(dev):
  """On-box re-measurement of the three peaks the roofline fractions are quoted against: a bare MFMA loop on random
  operands (ms_probe_peak: one wave per SIMD on every CU, operands in registers) in fp32 and bf16, and a 16-byte-per-lane copy of
  1 GiB.  HIP events on the current stream, best of 3 after a warm-up."""
  import ctypes
  import torch
  from mix_stage_amd import _lib
  L = _lib.lib()
  st = torch.cuda.current_stream(dev).cuda_stream
  scratch = torch.zeros(16, dtype=torch.float32, device=dev)
  n = 1 << 30
  src = torch.empty(n // 4, dtype=torch.float32, device=dev).normal_()
  dst = torch.empty_like(src)
  out = {}
  for name, kind, iters, a, b in (('fp32_mfma_tflops', 0, 4096, scratch, None), ('bf16_mfma_tflops', 1, 16384, scratch, None),
                                  ('copy_gbs', 2, n, src, dst)):
    best = 0.0
    for rep in range(4):
      work = ctypes.c_double(0)
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      _lib.check(L.ms_probe_peak(kind, iters, a.data_ptr(), b.data_ptr() if b is not None else None, ctypes.byref(work), st), 'ms_probe_peak')
      e1.record()
      torch.cuda.synchronize()
      if rep:
        best = max(best, work.value / (e0.elapsed_time(e1) * 1e-3))
    out[name] = round(best / (1e9 if kind == 2 else 1e12), 1)
  out['note'] = ('measured on this box by bench.py (ms_probe_peak): bare MFMA loops on random operands, one wave per SIMD; copy = read + written '
                 'bytes of a 1 GiB float4 copy.  The fractions above use the SPEC peaks (157.3 TF / 2500 TF / 8000 GB/s).')
  return out


def source_hash():
  """Hash of the kernel sources: PMC files record it so that a traffic figure is only quoted for the code that produced it."""
  h = hashlib.sha256()
  d = os.path.join(ROOT, 'mix_stage_amd', 'csrc')
  for f in sorted(os.listdir(d)):
    if f.endswith(('.hip', '.h', '.cpp')):
      h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
  return h.hexdigest()[:16]


def decoder_label_re(precision):
  # forward of decoder.1-3 (JL:69-77): grouped k3 conv, 256 -> 256 channels per group, M groups; with the batch statistics
  # (+bnstats: the normalising launch follows) or with the whole BatchNorm + LeakyReLU inside the launch (+bnfused)
  return re.compile(r'conv_fwd\S* k1x3 s1 Mg256 Kg768 g%d .*\+bn(stats|fused)' % M)


def block_and_segment(rows, n_g, precision):
  """Roofline entries of the north-star BLOCK (Conv1d + BatchNorm1d + LeakyReLU of a decoder layer, layers.py:77-78) and of
  SURVEY section 8(d)'s unit, the decoder segment decoder.0-3 + logits + softmax mixture (JL:69-83,190-194), from the same
  HIP-event timings: every launch that belongs to them, summed per G-step."""
  bt = B_PER_GPU * T
  esz = 2.0 if precision == 'bf16' else 4.0
  mode_peak = BF16_MFMA_PEAK_TFLOPS if precision == 'bf16' else (BF16_MFMA_PEAK_TFLOPS / 6.0 if precision == 'bf16x6' else FP32_MFMA_PEAK_TFLOPS)

  def avg(pattern):
    """(average launch duration in us over the matching labels, their labels) -- the same kernel also runs in D-steps (the
    generator's eval forward), so per-G-step sums use the launch counts of the model, not the counts of the timed steps"""
    pat = re.compile(pattern)
    hit = [r for r in rows if pat.search(r['label'])]
    n = sum(r['count'] for r in hit)
    return (sum(r['total_ms'] for r in hit) * 1e3 / n if n else 0.0), [r['label'].split('|')[-1] for r in hit]

  def entry(us, flops, nbytes, what, parts):
    t_mfma, t_hbm = flops / (mode_peak * 1e12), nbytes / (HBM_PEAK_GBS * 1e9)
    bound = 'hbm' if t_hbm >= t_mfma else 'mfma'
    s_ = us * 1e-6
    return dict(what=what, us=round(us, 2), bound=bound, algorithmic_flops=round(flops), algorithmic_bytes=round(nbytes),
                frac_hbm=round(nbytes / s_ / (HBM_PEAK_GBS * 1e9), 4), frac_mfma=round(flops / s_ / (mode_peak * 1e12), 4),
                frac=round(max(t_mfma, t_hbm) / s_, 4), time_lower_bound_us=round(max(t_mfma, t_hbm) * 1e6, 2), launches=parts)

  C = 256 * M
  conv_us, conv_l = avg(r'conv_fwd\S* k1x3 s1 Mg256 Kg768 g%d .*\+bn(stats|fused)' % M)
  bn_us, bn_l = avg(r'bn_finalize_apply\S* C%d N%d ' % (C, bt))     # the normalising launch, when BatchNorm is its own launch
  if not conv_l:
    return None, None
  blk_us = conv_us + bn_us
  blk_flops = 2.0 * 256 * 256 * 3 * bt * M
  blk_bytes = esz * (2.0 * bt * C + M * 256.0 * 768.0)
  block = entry(blk_us, blk_flops, blk_bytes,
                'decoder.1-3 block = grouped Conv1d(k3) + BatchNorm1d(train) + LeakyReLU forward: %s' %
                ('one launch (BatchNorm inside the conv launch)' if not bn_l else 'conv+statistics launch, then the normalising launch'),
                conv_l + bn_l)
  d0_us, d0_l = avg(r'conv_fwd\S* k1x3 s1 Mg256 Kg(798|816) g%d .*\+bn(stats|fused)' % M)
  lg_us, lg_l = avg(r'conv_fwd\S* k1x1 s1 Mg104 Kg256 g%d ' % M)
  mx_us, mx_l = avg(r'ew_softmax_mix_fwd')
  seg_us = 3 * conv_us + d0_us + 4 * bn_us + lg_us + mx_us        # launches per G-step: decoder.1-3, decoder.0, 4 x BN, logits, mixture
  # 26.89 GFLOP and 163.6 MB (fp32) / 81.8 MB (16-bit) at B=32, T=64, M=8; linear in B*T and M
  scale = (bt / 2048.0) * (M / 8.0)
  segment = entry(seg_us, 26.89e9 * scale, (81.8e6 if esz == 2.0 else 163.6e6) * scale,
                  'decoder segment = decoder.0-3 (+BatchNorm, LeakyReLU) + logits + softmax mixture, forward, per G-step '
                  '(SURVEY.md 8(d) unit): 3 x decoder.1-3 + decoder.0 + %d normalising launches + logits + mixture' % (4 if bn_l else 0),
                  conv_l + d0_l + bn_l + lg_l + mx_l)
  return block, segment


def chain_roofline(chain, rows, precision):
  """Roofline entry when the north-star unit runs as ONE launch (ms_decoder_chain_fwd: decoder.0-3 + logits + softmax mixture,
  SURVEY 8(d)'s unit): the train-mode launch of the G-steps (BatchNorm meetings, y_raw / y / z written for the backward pass) is the
  entry; the eval-mode launch of the D-steps (gan.py:106-110) is listed beside it."""
  native16 = precision in ('bf16', 'fp16')
  peak_tf = BF16_MFMA_PEAK_TFLOPS if native16 else FP32_MFMA_PEAK_TFLOPS

  def entry(r):
    avg_s = r['total_ms'] / r['count'] * 1e-3
    tf, gbs = r['flops'] / avg_s / 1e12, r['bytes'] / avg_s / 1e9
    t_mfma, t_hbm = r['flops'] / (peak_tf * 1e12), r['bytes'] / (HBM_PEAK_GBS * 1e9)
    bound = 'hbm' if t_hbm >= t_mfma else 'mfma'
    return dict(bound=bound, achieved=round(gbs if bound == 'hbm' else tf, 2), peak=HBM_PEAK_GBS if bound == 'hbm' else round(peak_tf, 1),
                unit='GB/s' if bound == 'hbm' else 'TFLOP/s', frac=round(max(t_mfma, t_hbm) / avg_s, 4), traffic=None,
                kernel=r['label'].split('|')[0], label=r['label'].split('|')[-1], avg_us=round(avg_s * 1e6, 2), launches=r['count'],
                algorithmic_flops_per_launch=round(r['flops']), algorithmic_bytes_per_launch=round(r['bytes']),
                frac_mfma=round(tf / peak_tf, 4), frac_hbm=round(gbs / HBM_PEAK_GBS, 4), achieved_tflops=round(tf, 2),
                achieved_gbs=round(gbs, 1), time_lower_bound_us=round(max(t_mfma, t_hbm) * 1e6, 2))

  train = [r for r in chain if r['label'].rstrip().endswith('train')]
  evalr = [r for r in chain if r['label'].rstrip().endswith('eval')]
  roof = entry((train or evalr)[0])
  roof['note'] = ('north-star unit as ONE launch: decoder.0-3 (grouped Conv1d k3 + BatchNorm1d + LeakyReLU) + logits + softmax mixture '
                  '(JL:69-83,106-115,186-194), M=%d groups, B*T = %d pixels; a workgroup carries one clip of one sub-generator through all blocks '
                  '(activations resident in LDS, weights streamed straight into registers); HIP events on the launch stream over 2 G-steps + 2 '
                  'D-steps (eager); algorithmic flops / bytes per SURVEY 8(d)' % (M, B_PER_GPU * T))
  seg = dict(roof)
  seg['what'] = 'decoder segment = the same single launch (SURVEY.md 8(d) unit), train mode, per G-step'
  seg['us'] = roof['avg_us']
  seg['launches'] = [roof['label']]
  roof['decoder_segment'] = seg
  roof['block'] = None        # the Conv+BN+LeakyReLU block is no launch of its own any more: a quarter of the chain's K loops + one meeting
  if train and evalr:
    roof['eval_launch'] = {k: v for k, v in entry(evalr[0]).items() if k in ('label', 'avg_us', 'frac', 'achieved_tflops', 'achieved_gbs', 'launches')}
  prep = [r for r in rows if 'chain_prep' in r['label']]
  if prep:
    roof['weight_stream_prep_us'] = round(prep[0]['total_ms'] / prep[0]['count'] * 1e3, 2)
  # HBM traffic of the launch from the committed PMC passes (tools/pmc_chain.sh: rocprofv3 cannot run inside bench.py): the newest
  # profiles/rNN_pmc_decoder.json whose `src_hash` equals the hash of the kernel sources being benchmarked, else null
  import glob
  for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_decoder.json')), reverse=True):
    try:
      pmc = json.load(open(path))
      ent = pmc.get(precision if M == 8 else '%s_m%d' % (precision, M))      # (the passes cover the headline and configs[1]: M = 8 / 4)
      if ent and 'hbm_bytes_per_launch' in ent and ent.get('src_hash') == source_hash() and B_PER_GPU * T == 2048:
        roof['traffic'] = ent['hbm_bytes_per_launch']
        roof['traffic_over_algorithmic'] = round(ent['hbm_bytes_per_launch'] / roof['algorithmic_bytes_per_launch'], 3)
        roof['traffic_note'] = 'HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, profiles/%s (same sources: %s)' % (os.path.basename(path), ent['src_hash'])
        break
    except (OSError, ValueError, KeyError):
      continue
  return roof


def kernel_roofline(ts, batch, kinds, precision):
  """Per-kernel launch durations from HIP events recorded on the launch stream by the library itself
  (ms_timing_*), over eager replays of the same steps as the timed region.  The roofline entry is the north-star kernel
  BY LABEL: the grouped decoder block forward (+ BN statistics), one layer = one launch."""
  import torch
  from mix_stage_amd import ops
  ops.timing_enable(True)
  saved = ts.use_graphs
  ts.use_graphs = False
  try:
    for k in kinds:
      ts.step(*batch, kind=k)
    torch.cuda.synchronize()
    rows = ops.timing_report()
  finally:
    ops.timing_enable(False)
    ts.use_graphs = saved
  if not rows:
    return None, []
  rows.sort(key=lambda r: -r['total_ms'])
  chain = [r for r in rows if 'decoder_chain_fwd' in r['label']]
  if chain:
    return chain_roofline(chain, rows, precision), rows
  pat = decoder_label_re(precision)
  dec = [r for r in rows if pat.search(r['label'])]
  if not dec:
    return None, rows
  r = dec[0]
  avg_s = r['total_ms'] / r['count'] * 1e-3
  native16 = precision == 'bf16'
  peak_tf = BF16_MFMA_PEAK_TFLOPS if native16 else (BF16_MFMA_PEAK_TFLOPS / 6.0 if 'patch6' in r['label'] else FP32_MFMA_PEAK_TFLOPS)
  tf = r['flops'] / avg_s / 1e12
  gbs = r['bytes'] / avg_s / 1e9
  t_mfma, t_hbm = r['flops'] / (peak_tf * 1e12), r['bytes'] / (HBM_PEAK_GBS * 1e9)
  bound = 'hbm' if t_hbm >= t_mfma else 'mfma'
  roof = dict(bound=bound,
              achieved=round(gbs if bound == 'hbm' else tf, 2), peak=HBM_PEAK_GBS if bound == 'hbm' else round(peak_tf, 1),
              unit='GB/s' if bound == 'hbm' else 'TFLOP/s',
              frac=round((gbs / HBM_PEAK_GBS) if bound == 'hbm' else (tf / peak_tf), 4), traffic=None,
              kernel=r['label'].split('|')[0], label=r['label'].split('|')[-1], avg_us=round(avg_s * 1e6, 2), launches=r['count'],
              algorithmic_flops_per_launch=round(r['flops']), algorithmic_bytes_per_launch=round(r['bytes']),
              frac_mfma=round(tf / peak_tf, 4), frac_hbm=round(gbs / HBM_PEAK_GBS, 4),
              achieved_tflops=round(tf, 2), achieved_gbs=round(gbs, 1),
              time_lower_bound_us=round(max(t_mfma, t_hbm) * 1e6, 2),
              frac_of_lower_bound=round(max(t_mfma, t_hbm) / avg_s, 4),
              note='north-star kernel by label: grouped decoder block forward, BatchNorm statistics or the whole BatchNorm + LeakyReLU inside the launch (decoder.1-3: k3, %d groups, '
                   '256->256 channels per group, B*T = %d pixels), one launch per layer; HIP events on the launch stream over '
                   '2 G-steps + 2 D-steps (eager); bound = the larger of flops/peak and bytes/HBM-peak' % (M, B_PER_GPU * T))
  n_g = sum(1 for k in kinds if k == 'G')
  try:
    roof['block'], roof['decoder_segment'] = block_and_segment(rows, max(1, n_g), precision)
  except Exception as e:  # noqa: BLE001
    roof['block'] = roof['decoder_segment'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
  # HBM traffic of that launch from the committed PMC passes (tools/pmc_decoder.sh; rocprofv3 cannot run inside bench.py):
  # quoted only when the file was produced by these very kernel sources
  try:
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r03_pmc_decoder.json')))
    ent = pmc.get(precision)
    # (the passes measure the HEADLINE decoder layer: 8 groups, 2048 pixels -- other configs get no traffic figure)
    if ent and ent.get('src_hash') == source_hash() and M == 8 and B_PER_GPU * T == 2048:
      roof['traffic'] = ent['hbm_bytes_per_launch']
      if isinstance(roof.get('block'), dict) and 'block_hbm_bytes' in ent:
        roof['block']['traffic'] = ent['block_hbm_bytes']
        roof['block']['traffic_over_algorithmic'] = round(ent['block_hbm_bytes'] / roof['block']['algorithmic_bytes'], 3)
      roof['traffic_note'] = 'HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, profiles/r03_pmc_decoder.json (same sources: %s)' % ent['src_hash']
  except (OSError, ValueError, KeyError):
    pass
  return roof, rows


def bf16_extra(dev, batch, args):
  """The same workload in the native bf16 arithmetic mode (BASELINE configs[1]/[3] dtype: bf16 operands and activations, fp32
  accumulate / statistics / master weights), measured after the fp32 headline in the same process: G-step and D-step times,
  their blend, and the north-star kernel's roofline entry.  Reported next to the headline, never as `value` (the 1e-4 pose
  bar belongs to the fp32 path; bf16 pose L1 vs the fp64 oracle is measured in tests/test_gpu_model16.py)."""
  import torch
  from mix_stage_amd.train_step import MixStageTrainStep
  model = build_model(dev, 'bf16')
  ts = MixStageTrainStep(model, use_graphs=not args.no_graphs, time_steps=T)
  torch.manual_seed(args.seed)
  res = {}
  n_k = max(5, min(20, args.steps))
  for kind in ('G', 'D'):
    for _ in range(3):
      ts.step(*batch, kind=kind)
    res[kind] = time_steps(ts, batch, n_k, kind, 1, None, dev)[0] / n_k
  g_ms, d_ms = 1e3 * res['G'], 1e3 * res['D']
  out = dict(dtype='bf16 (fp32 accumulate, fp32 BN statistics, fp32 master weights)', g_step_ms=round(g_ms, 4), d_step_ms=round(d_ms, 4),
             value_blend_50_50=round(B_PER_GPU / (0.5e-3 * (g_ms + d_ms)), 2), unit='clips/s',
             losses_finite=all(float(l.detach()) == float(l.detach()) for l in ts.losses))
  if not args.no_kernel_timing:
    out['roofline'], _ = kernel_roofline(ts, batch, ['G', 'D', 'G', 'D'], 'bf16')
  del ts, model
  torch.cuda.empty_cache()
  return out


def time_steps(ts, batch, n, kind, world, dist, dev):
  import torch
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  kinds = [ts.step(*batch, kind=kind) for _ in range(n)]
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
  return elapsed, kinds


def bench_inference(args):
  """configs[4]: the generator's eval forward (sample_flag=1: style ids given, BatchNorm from the running statistics folded into
  the prepared fp16 weights) on B=1024 clips, captured in a HIP graph; a step = one replay."""
  import torch
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  model = build_model(dev, 'fp32')
  A.set_compute_dtype(model, 'fp16')
  model.eval()
  A.set_inference_folding(model, True)
  audio, pose, labels, style = O.synthetic_batch(B_PER_GPU, T=T, F_=F_MEL, P=P, M=M, S=S, seed=1234)
  style = (style + 3) % S                           # transfer to another speaker's style (trainer.py:1367-1386)
  st = [t.to(dev) for t in (audio, labels, pose, style)]
  kw = O.model_kwargs(st[3], T); kw['sample_flag'] = 1
  with torch.no_grad():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      model([st[0], st[1]], st[2], **kw)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
      y_cap, _, _ = model([st[0], st[1]], st[2], **kw)
    for _ in range(args.warmup):
      g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      g.replay()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
  fwd_gflop = (B_PER_GPU * T) / 2048.0 * 99.0
  ms = 1e3 * elapsed / args.steps
  out = {'metric': CONFIGS[args.config]['metric'], 'value': round(B_PER_GPU * args.steps / elapsed, 2), 'unit': 'clips/s', 'n_gpus': 1,
         'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True, 'scaling': 'weak',
         'vs_baseline': None, 'dtype': 'f16 (fp32 accumulate; eval BatchNorm folded into the prepared weights)', 'data': 'synthetic',
         'config': {'workload': 'Mix-StAGE generator eval forward, style transfer (sample_flag=1), B=%d clips, T=%d, M=S=%d, HIP-graph replay'
                                % (B_PER_GPU, T, M), 'global_batch': B_PER_GPU, 'parallelism': 'dp1', 'hip_graphs': True},
         'output_finite': bool(torch.isfinite(y_cap).all()),
         'roofline': {'bound': 'mfma', 'achieved': round(fwd_gflop / ms, 2), 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                      'frac': round(fwd_gflop / ms / BF16_MFMA_PEAK_TFLOPS, 4), 'traffic': None,
                      'note': 'whole forward: %.0f algorithmic GFLOP per replay (SURVEY A.3) / replay time, against the dense fp16 MFMA peak' % fwd_gflop},
         'cpu_baseline': None}
  print(json.dumps(out))


def main():
  args = parse()
  if CONFIGS[args.config]['kind'] == 'infer':
    return bench_inference(args)
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and 'RANK' not in os.environ:
    sys.exit(self_launch(args))
  if world != args.gpus:
    raise SystemExit('WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
  if world > 1 and not args.worker and os.environ.get('MS_BENCH_SUPERVISE', '1') != '0':
    sys.exit(supervise(args))
  if os.environ.get('MS_BENCH_NO_GRAPHS') == '1':
    args.no_graphs = True
  wd = Watchdog()
  wd.beat('import', 900)
  import torch
  import torch.distributed as dist
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if args.precision == 'bf16x6':
    os.environ['MS_PRECISION'] = 'bf16x6'       # read when the library is loaded
  if args.same_device:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  # MS_DP_SINGLE_RANK=1 under a launcher: ONE rank takes the data-parallel form of the step (split graphs, eager RCCL
  # all-reduce) -- what that machinery costs without any inter-GPU transfer, measurable on a one-GPU box
  if world > 1 or (os.environ.get('MS_DP_SINGLE_RANK') == '1' and 'RANK' in os.environ):
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.dist_backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(args.dist_backend)

  from oracle import mixstage_oracle as O
  from mix_stage_amd.train_step import MixStageTrainStep
  model = build_model(dev, args.precision)
  ts = MixStageTrainStep(model, use_graphs=not args.no_graphs, time_steps=T, bn_sync=args.bn_sync,
                         grad_buckets=int(os.environ['MS_GRAD_BUCKETS']) if os.environ.get('MS_GRAD_BUCKETS') else None,
                         overlap_allreduce=os.environ.get('MS_OVERLAP_ALLREDUCE', '0') == '1',
                         grad_exchange=os.environ.get('MS_GRAD_EXCHANGE', 'fp32'))
  # every rank gets its own shard of synthetic clips (pure data parallel, weak scaling: 32 clips per GPU)
  audio, pose, labels, style = O.synthetic_batch(B_PER_GPU, T=T, F_=F_MEL, P=P, M=M, S=S, seed=1234 + rank)
  batch = [t.to(dev) for t in (audio, labels, pose, style)]
  torch.manual_seed(args.seed)          # identical host generators on all ranks -> identical D/G decisions

  wd.beat('warmup', 300)                # (the first steps capture the graphs and set up RCCL's channels)
  for _ in range(args.warmup):
    ts.step(*batch)
    torch.cuda.synchronize()
    wd.beat('warmup', 300)
  wd.beat('timed', 120)
  elapsed, kinds = time_steps(ts, batch, args.steps, None, world, dist, dev)
  wd.beat('per-kind', 120)
  losses = [float(l.detach()) for l in ts.losses]
  finite = all(l == l and abs(l) < 1e6 for l in losses)
  per_kind = None
  if not args.no_per_kind:
    n_k = max(5, min(20, args.steps))
    per_kind = {}
    for kind in ('G', 'D'):
      for _ in range(2):
        ts.step(*batch, kind=kind)
      per_kind[kind] = time_steps(ts, batch, n_k, kind, world, dist, dev)[0] / n_k
      wd.beat('per-kind', 120)
  ts.check_health()                    # raises if a launch whose workgroups meet in-launch timed out (outputs would be NaN)
  wd.beat('report', 1200)

  out = None
  if rank == 0:
    n_g = sum(k == 'G' for k in kinds)
    dtype = {'fp32': 'f32', 'bf16x6': 'f32 via bf16x6 (exact 3-way bf16 split of both operands, 6 of 9 products, fp32 accumulate)',
             'bf16': 'bf16 (fp32 accumulate, fp32 BN statistics, fp32 master weights)'}[args.precision]
    out = {
        'metric': CONFIGS[args.config]['metric'], 'value': round(world * B_PER_GPU * args.steps / elapsed, 2),
        'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
        'config': {'workload': 'Mix-StAGE GAN train step (reference D/G coin flip, seed %d: %d G + %d D steps), '
                               'B=%d clips per GPU, T=%d, %d-mel, %d-dim pose, M=S=%d, audio branch pinned; %s'
                               % (args.seed, n_g, args.steps - n_g, B_PER_GPU, T, F_MEL, P, M,
                                  {'fp32': 'fp32 arithmetic (the reference trains in fp64, trainer.py:138; held to it at 1e-4 pose L1)',
                                   'bf16x6': 'fp32 products from bf16 splits (the reference trains in fp64)',
                                   'bf16': 'bf16 arithmetic, fp32 accumulation and master weights (the reference trains in fp64)'}[args.precision]),
                   'global_batch': world * B_PER_GPU, 'parallelism': 'dp%d' % world, 'hip_graphs': not args.no_graphs,
                   'bn_sync': args.bn_sync},
        'last_losses': [round(l, 5) for l in losses], 'losses_finite': finite,
    }
    if world > 1:
      out['dp_fallback'] = os.environ.get('MS_DP_FALLBACK') or False
      coll = 'RCCL' if args.dist_backend == 'nccl' else args.dist_backend
      out['config']['grad_exchange'] = 'captured in the step graph' if ts.capture_allreduce else ('eager %s all-reduce between two graphs' % coll if not args.no_graphs else 'eager %s all-reduce' % coll)
    if per_kind:
      g_ms, d_ms = 1e3 * per_kind['G'], 1e3 * per_kind['D']
      peak = {'fp32': FP32_MFMA_PEAK_TFLOPS, 'bf16x6': FP32_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS}[args.precision]
      out.update(g_step_ms=round(g_ms, 4), d_step_ms=round(d_ms, 4),
                 value_blend_50_50=round(world * B_PER_GPU / (0.5e-3 * (g_ms + d_ms)), 2),
                 whole_step={'g_step_tflops': round(G_STEP_GFLOP / g_ms, 2), 'd_step_tflops': round(D_STEP_GFLOP / d_ms, 2),
                             'mfma_frac_g_step': round(G_STEP_GFLOP / g_ms / peak, 4),
                             'mfma_frac_d_step': round(D_STEP_GFLOP / d_ms / peak, 4),
                             'note': 'per-rank algorithmic GFLOP of a whole step (G: 3 x 99.0, D: 99.0 + 6 x 0.215; SURVEY A.3) '
                                     '/ measured step time, against the matrix peak of the arithmetic mode (%.1f TF)' % peak})
  if rank == 0 and world > 1:     # measured at N=1 only (per-kernel events / the CPU oracle would distort the ranks' lockstep)
    out['roofline'] = None
    out['cpu_baseline'] = None
  if rank == 0 and world == 1:
    roof, rows = (None, [])
    if not args.no_kernel_timing:
      try:
        roof, rows = kernel_roofline(ts, batch, ['G', 'D', 'G', 'D'], args.precision)
      except Exception as e:  # noqa: BLE001
        roof = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    out['roofline'] = roof
    if isinstance(roof, dict) and 'error' not in roof:
      try:
        roof['measured_peaks'] = measured_peaks(dev)
        mp = roof['measured_peaks']
        meas = mp['bf16_mfma_tflops'] if args.precision == 'bf16' else mp['fp32_mfma_tflops']
        if roof.get('achieved_tflops') and meas:
          roof['frac_of_measured_mfma_peak'] = round(roof['achieved_tflops'] / meas, 4)
      except Exception as e:  # noqa: BLE001
        roof['measured_peaks'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    out['kernel_table'] = [dict(label=r['label'].split('|')[-1], count=r['count'], avg_us=round(1e3 * r['total_ms'] / r['count'], 2),
                                total_ms=round(r['total_ms'], 3),
                                tflops=round(r['flops'] / (r['total_ms'] / r['count'] * 1e-3) / 1e12, 2))
                           for r in rows[:12]]
    # auxiliary measurements must never cost the headline line: a failure is reported in place
    if args.precision == 'fp32' and not args.no_bf16_extra and not args.no_per_kind:
      try:
        out['bf16'] = bf16_extra(dev, batch, args)
        b16 = out['bf16']
        out['bf16_value_blend_50_50'] = b16.get('value_blend_50_50')
        out['bf16_g_step_ms'], out['bf16_d_step_ms'] = b16.get('g_step_ms'), b16.get('d_step_ms')
        blk = (b16.get('roofline') or {}).get('block') or {}
        out['bf16_decoder_block_us'], out['bf16_decoder_block_frac'] = blk.get('us'), blk.get('frac')
        seg16 = (b16.get('roofline') or {}).get('decoder_segment') or {}
        out['bf16_decoder_segment_us'], out['bf16_decoder_segment_frac'] = seg16.get('us'), seg16.get('frac')
      except Exception as e:  # noqa: BLE001
        out['bf16'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
    try:
      out['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(args.seed)
    except Exception as e:  # noqa: BLE001
      out['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
  if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    print(json.dumps(out))


if __name__ == '__main__':
  main()
