#!/usr/bin/env python3
"""Headline benchmark: Mix-StAGE train-step clips/sec (B=32 per GPU, T=64, 128-mel, 104-dim pose, M=S=8), fp32.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one reference train step (trainer.py:604-674 contract, see mix_stage_amd/train_step.py): zero_grad ->
GAN.forward (D-step or G-step by the reference's seeded host coin flip, gan.py:105) -> backward -> gradient all-reduce
(N>1) -> clip_grad_norm_(.,1) -> Adam(1e-4).  Inputs are synthetic, resident in HBM before the timed region; weights are
the name-keyed deterministic fill.  The curriculum is pinned to the audio branch (thresh = 1), as stated in
BASELINE.md.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

B_PER_GPU, T, F_MEL, P, M, S = 32, 64, 128, 104, 8, 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=60)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--no-graphs', action='store_true', help='eager launches instead of HIP-graph replay')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-kernel-timing', action='store_true')
  ap.add_argument('--seed', type=int, default=4321)
  ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x6'],
                  help='fp32: exact fp32 matrix products (default, the headline); bf16x6: both operands split exactly into 3 bf16 '
                       'parts, 6 of 9 partial products on the bf16 pipe, fp32 accumulate (same measured accuracy)')
  ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI); gloo only for smoke-testing the DP path')
  ap.add_argument('--same-device', action='store_true', help='smoke test: all ranks share cuda:0 (needs --dist-backend gloo)')
  return ap.parse_args()


def build_model(dev):
  import torch
  import mix_stage_amd as A
  from oracle import mixstage_oracle as O     # deterministic weight fill + synthetic inputs (shared with the tests)
  G = A.JointLateClusterSoftStyle4_G(time_steps=T, out_feats=P, num_clusters=M, style_dict={i: i for i in range(S)},
                                     style_dim=10, lambda_id=0.1, argmax=1, some_grad_flag=1, train_only=1, shape={})
  D = A.Speech2Gesture_D(in_channels=P)
  model = A.GAN(G, D, criterion='L1Loss', input_modalities=['audio/log_mel_400'], update_D_prob_flag=0, no_grad=0)
  model.load_state_dict(O.deterministic_state(model.state_dict()))
  model.G.thresh.value, model.G.thresh.iters = 1, 10 ** 9
  return model.to(dev)


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
  """The oracle (pure PyTorch on the host cores, fp32) on a bounded sample of the same workload: G-steps and D-steps
  at B=32 after one untimed step of each kind (2 timed steps per kind, 1 if a step takes longer than 8 s)."""
  import torch
  from oracle import mixstage_oracle as O
  cores = min(usable_cores(), 32)       # oversubscribed intra-op threads make PyTorch CPU convs much slower
  torch.set_num_threads(cores)
  model = O.build_gan(M=M, S=S, T=T, P=P)
  og = torch.optim.Adam(model.G.parameters(), lr=1e-4)
  od = torch.optim.Adam(model.D.parameters(), lr=1e-4)
  audio, pose, labels, style = O.synthetic_batch(B_PER_GPU, T=T, F_=F_MEL, P=P, M=M, S=S, seed=1234)
  times, reps = {}, {}
  for kind in ('G', 'D'):
    t0 = time.perf_counter()
    O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
    warm = time.perf_counter() - t0
    n = 2 if warm < 8.0 else 1
    t0 = time.perf_counter()
    for _ in range(n):
      O.oracle_train_step(model, og, od, audio, pose, labels, style, kind, T=T)
    times[kind], reps[kind] = (time.perf_counter() - t0) / n, n
  blended = 0.5 * (times['G'] + times['D'])       # D_prob = 0.5 (gan.py:27)
  return dict(value=round(B_PER_GPU / blended, 2), unit='clips/s', cores=cores, kind='port',
              sample='oracle (PyTorch CPU fp32, %d threads): %d G-steps + %d D-steps at B=32 after 1 warm-up each; '
                     'G %.3f s, D %.3f s per step, 50/50 blend' % (cores, reps['G'], reps['D'], times['G'], times['D']))


def kernel_roofline(ts, batch, kinds):
  """Per-kernel launch durations from HIP events recorded on the launch stream by the library itself
  (ms_timing_*), over eager replays of the same steps as the timed region."""
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
  # aggregate by kernel symbol (what rocprofv3 --stats reports); launches without a symbol keep their label
  by_sym = {}
  for r in rows:
    sym = r['label'].split('|')[0]
    a = by_sym.setdefault(sym, dict(sym=sym, count=0, total_ms=0.0, flops=0.0, bytes=0.0))
    a['count'] += r['count']; a['total_ms'] += r['total_ms']
    a['flops'] += r['flops'] * r['count']; a['bytes'] += r['bytes'] * r['count']
  syms = sorted(by_sym.values(), key=lambda a: -a['total_ms'])
  top = syms[0]
  avg_s = top['total_ms'] / top['count'] * 1e-3
  achieved = top['flops'] / (top['total_ms'] * 1e-3) / 1e12
  # bf16x6 kernels do 6 bf16 MFMA products per algorithmic fp32 product: their ceiling in algorithmic flops is 1/6 of the dense
  # bf16 peak (MI355X_MICROARCH.md: 2.5 PFLOP/s)
  peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if 'patch6' in top['sym'] else FP32_MFMA_PEAK_TFLOPS
  roof = dict(bound='mfma', achieved=round(achieved, 2), peak=round(peak, 1), unit='TFLOP/s',
              frac=round(achieved / peak, 4), traffic=None, kernel=top['sym'],
              avg_us=round(avg_s * 1e6, 2), launches=top['count'],
              algorithmic_flops_per_launch=round(top['flops'] / top['count']),
              algorithmic_bytes_per_launch=round(top['bytes'] / top['count']),
              note='dominant kernel symbol by total time over 2 G-steps + 2 D-steps; avg over all its launches (all layer '
                   'shapes), HIP events on the launch stream; same aggregation as rocprofv3 --kernel-trace --stats')
  # HBM traffic of that kernel from the committed PMC passes (tools/pmc_summary.py; rocprofv3 cannot run inside bench.py)
  try:
    pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))['kernels'].get(top['sym'])
    if pmc:
      roof['traffic'] = pmc['hbm_bytes_per_launch']
      roof['traffic_note'] = 'HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from profiles/r01_pmc_traffic.json'
  except (OSError, ValueError, KeyError):
    pass
  rows.sort(key=lambda r: -r['total_ms'])
  return roof, rows


def main():
  args = parse()
  import torch
  import torch.distributed as dist
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr '
                       '127.0.0.1 bench.py --gpus %d ...' % (args.gpus, args.gpus))
  if args.precision == 'bf16x6':
    os.environ['MS_PRECISION'] = 'bf16x6'       # read when the library is loaded
  if args.same_device:
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if args.dist_backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)
    else:
      dist.init_process_group(args.dist_backend)

  from oracle import mixstage_oracle as O
  from mix_stage_amd.train_step import MixStageTrainStep
  model = build_model(dev)
  ts = MixStageTrainStep(model, use_graphs=not args.no_graphs, time_steps=T)
  # every rank gets its own shard of synthetic clips (pure data parallel, weak scaling: 32 clips per GPU)
  audio, pose, labels, style = O.synthetic_batch(B_PER_GPU, T=T, F_=F_MEL, P=P, M=M, S=S, seed=1234 + rank)
  batch = [t.to(dev) for t in (audio, labels, pose, style)]
  torch.manual_seed(args.seed)          # identical host generators on all ranks -> identical D/G decisions

  kinds = []
  for _ in range(args.warmup):
    ts.step(*batch)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    kinds.append(ts.step(*batch))
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  if world > 1:
    tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
  losses = [float(l.detach()) for l in ts.losses]
  finite = all(l == l and abs(l) < 1e6 for l in losses)

  out = None
  if rank == 0:
    n_g = sum(k == 'G' for k in kinds)
    out = {
        'metric': 'train-step clips/sec (B=32, T=64, M=8)', 'value': round(world * B_PER_GPU * args.steps / elapsed, 2),
        'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * elapsed / args.steps, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if args.precision == 'fp32' else 'f32 via bf16x6 (exact 3-way bf16 split of both operands, 6 of 9 products, fp32 accumulate)', 'data': 'synthetic',
        'config': {'workload': 'Mix-StAGE GAN train step (reference D/G coin flip, seed %d: %d G + %d D steps), '
                               'B=%d clips per GPU, T=%d, %d-mel, %d-dim pose, M=S=%d, audio branch pinned'
                               % (args.seed, n_g, args.steps - n_g, B_PER_GPU, T, F_MEL, P, M),
                   'global_batch': world * B_PER_GPU, 'parallelism': 'dp%d' % world, 'hip_graphs': not args.no_graphs,
                   'bn_sync': 'local'},
        'last_losses': [round(l, 5) for l in losses], 'losses_finite': finite,
    }
  if rank == 0 and world > 1:     # measured at N=1 only (per-kernel events / the CPU oracle would distort the ranks' lockstep)
    out['roofline'] = None
    out['cpu_baseline'] = None
  if rank == 0 and world == 1:
    roof, rows = (None, [])
    if not args.no_kernel_timing:
      roof, rows = kernel_roofline(ts, batch, ['G', 'D', 'G', 'D'])
    out['roofline'] = roof
    out['kernel_table'] = [dict(label=r['label'].split('|')[-1], count=r['count'], avg_us=round(1e3 * r['total_ms'] / r['count'], 2),
                                total_ms=round(r['total_ms'], 3),
                                tflops=round(r['flops'] / (r['total_ms'] / r['count'] * 1e-3) / 1e12, 2))
                           for r in rows[:12]]
    out['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(args.seed)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    print(json.dumps(out))


if __name__ == '__main__':
  main()
