"""Launch-by-launch list of ONE replayed residual step from a rocprofv3 kernel trace (tools/residual.sh): the trace ends with
the bare graph replays of tools/trace_residual.py, whose kernel sequence is periodic -- the period is one step.
usage: residual_summary.py kernel_trace.csv out.json KEY"""
import csv, json, os, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
K = None
for k in range(2, len(names) // 4):
  if names[-k:] == names[-2 * k:-k] == names[-3 * k:-2 * k] and len(set(names[-k:])) > 1:
    K = k
    break
assert K, 'no period found'
step = rows[-2 * K:-K]
nxt = rows[-K]
t0 = int(step[0]['Start_Timestamp'])
out, prev_end = [], t0
for r in step:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  nm = re.sub(r'^void ', '', r['Kernel_Name'])
  nm = re.sub(r'\(.*$', '', nm)[:100]
  out.append(dict(k=nm, t=round((s - t0) / 1e3, 2), d=round((e - s) / 1e3, 2), gap=round((s - prev_end) / 1e3, 2)))
  prev_end = max(prev_end, e)
period = (int(nxt['Start_Timestamp']) - t0) / 1e3
res = dict(launches=len(out), busy_us=round(sum(k['d'] for k in out), 1), gaps_inside_us=round(sum(k['gap'] for k in out), 1),
           step_period_us=round(period, 1), between_steps_us=round(period - (prev_end - t0) / 1e3, 1), kernels=out)
path, key = sys.argv[2], sys.argv[3]
allr = json.load(open(path)) if os.path.exists(path) else {}
allr[key] = res
json.dump(allr, open(path, 'w'), indent=1)
print(key, {k: v for k, v in res.items() if k != 'kernels'})
for k in out: print('   %8.1f us  +%6.1f gap  %s' % (k['d'], k['gap'], k['k']))
