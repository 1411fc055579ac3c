"""Timeline of the LAST replayed step from a rocprofv3 kernel trace: per kernel, in start order, the start offset from the
step's first kernel, the duration and the gap to the previous kernel's end.  A step ends with its last `adam` kernel.
usage: trace_summary.py kernel_trace.csv out.json"""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# step boundaries: an adam kernel followed by a non-adam kernel
ends = [i for i, n in enumerate(names) if 'adam' in n.lower() and (i + 1 == len(names) or 'adam' not in names[i + 1].lower())]
last, prev = ends[-1], ends[-2]
step = rows[prev + 1:last + 1]
t0 = int(step[0]['Start_Timestamp'])
out, prev_end = [], t0
for r in step:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  nm = re.sub(r'^void ', '', r['Kernel_Name'])
  nm = re.sub(r'\(.*$', '', nm)[:110]
  out.append(dict(k=nm, t=round((s - t0) / 1e3, 2), d=round((e - s) / 1e3, 2), gap=round((s - prev_end) / 1e3, 2),
                  wg=int(r.get('Grid_Size', 0) or 0) // max(1, int(r.get('Workgroup_Size', 1) or 1))))
  prev_end = max(prev_end, e)
json.dump(dict(total_us=round((prev_end - t0) / 1e3, 2), n=len(out), kernels=out), open(sys.argv[2], 'w'))
print('step: %d kernels, %.1f us; busy %.1f us, gaps %.1f us' % (len(out), (prev_end - t0) / 1e3, sum(k['d'] for k in out), sum(k['gap'] for k in out)))
