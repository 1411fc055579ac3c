"""GPU: A/B of library builds on ONE box (boxes differ by several %): the captured G-step / D-step time of the bench workload,
each build in a fresh process, alternating, N rounds.
  python tools/ab_libs.py PRECISION ROUNDS name=path[,ENV=VAL...] name=path ...      (path '' = the in-tree build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, bench
from mix_stage_amd.train_step import MixStageTrainStep
from oracle import mixstage_oracle as O
dev = torch.device('cuda:0')
audio, pose, labels, style = O.synthetic_batch(32, M=8, S=8)
batch = [t.to(dev) for t in (audio, labels, pose, style)]
model = bench.build_model(dev, sys.argv[1])
ts = MixStageTrainStep(model, use_graphs=True)
out = []
for kind in 'GD':
  for _ in range(6):
    ts.step(*batch, kind=kind)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(60):
    ts.step(*batch, kind=kind)
  torch.cuda.synchronize()
  out.append((time.perf_counter() - t0) / 60 * 1e3)
print('RESULT %%.4f %%.4f' %% tuple(out))
''' % ROOT
precision, rounds = sys.argv[1], int(sys.argv[2])
cfgs = []
for a in sys.argv[3:]:
  name, rest = a.split('=', 1)
  parts = rest.split(',')
  env = dict(kv.split('=', 1) for kv in parts[1:])
  if parts[0]:
    env['MS_LIB_PATH'] = os.path.join(ROOT, parts[0])
  cfgs.append((name, env))
acc = {n: [] for n, _ in cfgs}
for r in range(rounds):
  for name, env in cfgs:
    e = dict(os.environ); e.update(env)
    o = subprocess.run([sys.executable, '-c', CHILD, precision], env=e, capture_output=True, text=True)
    line = [l for l in o.stdout.splitlines() if l.startswith('RESULT')]
    if not line:
      print(name, 'FAILED', o.stderr[-400:]); continue
    g, d = map(float, line[0].split()[1:])
    acc[name].append((g, d))
    print('round %d %-12s G %.3f D %.3f' % (r, name, g, d), flush=True)
for name, v in acc.items():
  if v:
    print('%-12s G %.3f  D %.3f  (mean of %d)' % (name, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), len(v)))
