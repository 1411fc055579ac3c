"""GPU: A/B of environment settings on ONE box -- bench.py in fresh processes, alternating, G / D step times and the rows of the
kernel table that match a filter.   usage: python tools/sweep_env.py [--prec fp32] [--filter wgrad] [--rounds 2] "A=1 B=2" "A=3" ...
('-' = no variables)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
prec, flt, rounds = 'fp32', 'wgrad', 2
while args and args[0].startswith('--'):
  k, v = args[0], args[1]; args = args[2:]
  if k == '--prec': prec = v
  elif k == '--filter': flt = v
  elif k == '--rounds': rounds = int(v)
res = {a: [] for a in args}
for r in range(rounds):
  for a in args:
    env = dict(os.environ)
    if a != '-':
      for kv in a.split():
        k, v = kv.split('='); env[k] = v
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-bf16-extra',
                          '--precision', prec], env=env, capture_output=True, text=True)
    try:
      j = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
      print(a, 'FAILED', out.stderr[-600:]); continue
    res[a].append((j['g_step_ms'], j['d_step_ms']))
    if r == 0:
      for row in j.get('kernel_table', []):
        if flt in row['label']:
          print('   [%s] %-70s x%d %8.1f us %6.1f TF' % (a, row['label'][:70], row['count'], row['avg_us'], row['tflops']))
for a in args:
  print('%-40s G %s  D %s' % (a, ' '.join('%.3f' % x[0] for x in res[a]), ' '.join('%.3f' % x[1] for x in res[a])))
