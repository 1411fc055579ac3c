"""Summarises tools/pmc_step.sh's passes: per kernel symbol of the step, per-launch averages of the SQ / TCC counters and
derived fractions (MFMA busy, wait / issue / stall split of the wave cycles, LDS conflicts, HBM bytes with the gfx950
correction (2*FETCH_SIZE + WRITE_SIZE) KiB, clock)."""
import collections, csv, glob, json, os, sys
root, outj, prec, kind = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
wgs = {}
for f in glob.glob(root + '/p*/*/*counter_collection.csv'):
  for r in csv.DictReader(open(f)):
    a = agg[r['Kernel_Name']][r['Counter_Name']]
    a[0] += 1; a[1] += float(r['Counter_Value'])
    try:
      wgs[r['Kernel_Name']] = int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))
    except Exception:
      pass
for f in glob.glob(root + '/p1/*/*kernel_trace.csv'):
  for r in csv.DictReader(open(f)):
    a = dur[r['Kernel_Name']]
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
out = {}
for k, cs in agg.items():
  if k not in dur:
    continue
  us = dur[k][1] / dur[k][0]
  w = {c: round(v / n, 1) for c, (n, v) in cs.items()}
  w['avg_us_under_pmc'] = round(us, 2)
  w['launches_seen'] = dur[k][0]
  der = {}
  if 'GRBM_GUI_ACTIVE' in w:
    der['clock_ghz_est'] = round(w['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3), 3)
  if 'SQ_VALU_MFMA_BUSY_CYCLES' in w:
    der['mfma_busy_frac_of_launch_at_2400'] = round(w['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * 2400.0), 4)
  if w.get('SQ_WAVE_CYCLES'):
    for k2, name in (('SQ_WAIT_ANY', 'wait_frac_of_wave_cycles'), ('SQ_ACTIVE_INST_ANY', 'issue_frac_of_wave_cycles'),
                     ('SQ_WAIT_INST_ANY', 'stall_frac_of_wave_cycles')):
      if k2 in w:
        der[name] = round(w[k2] / w['SQ_WAVE_CYCLES'], 4)
    if w.get('SQ_WAVES'):
      # quad-cycles -> us per wave at the estimated clock: how long the average wave lives compared with the launch
      clk = der.get('clock_ghz_est', 2.4)
      der['avg_wave_life_us'] = round(w['SQ_WAVE_CYCLES'] * 4 / w['SQ_WAVES'] / (clk * 1e3), 2)
  if w.get('SQ_WAVES'):
    for c, name in (('SQ_INSTS_VALU', 'valu_per_wave'), ('SQ_INSTS_SALU', 'salu_per_wave'), ('SQ_INSTS_LDS', 'lds_per_wave'),
                    ('SQ_INSTS_MFMA', 'mfma_per_wave'), ('SQ_INSTS_VMEM_RD', 'vmem_rd_per_wave'), ('SQ_INSTS_VMEM_WR', 'vmem_wr_per_wave')):
      if c in w:
        der[name] = round(w[c] / w['SQ_WAVES'], 1)
  if w.get('SQ_LDS_IDX_ACTIVE'):
    der['lds_conflict_frac'] = round(w.get('SQ_LDS_BANK_CONFLICT', 0) / w['SQ_LDS_IDX_ACTIVE'], 4)
  if 'FETCH_SIZE' in w and 'WRITE_SIZE' in w:
    der['hbm_mb_per_launch'] = round((2 * w['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024 / 1e6, 2)
    der['hbm_gbs'] = round((2 * w['FETCH_SIZE'] + w['WRITE_SIZE']) * 1024 / (us * 1e3), 1)
  if 'TCC_HIT_sum' in w:
    der['l2_hit_frac'] = round(w['TCC_HIT_sum'] / max(1.0, w['TCC_HIT_sum'] + w.get('TCC_MISS_sum', 0)), 4)
  w['derived'] = der
  w['total_us_per_step'] = round(dur[k][1] / max(1, dur[k][0]) * dur[k][0], 1)
  out[k[:110]] = w
res = {'precision': prec, 'kind': kind, 'note': 'eager steps under rocprofv3 --pmc (one counter group per run); durations from pass 1',
       'kernels': dict(sorted(out.items(), key=lambda kv: -kv[1]['avg_us_under_pmc'] * kv[1]['launches_seen']))}
os.makedirs(os.path.dirname(outj), exist_ok=True)
json.dump(res, open(outj, 'w'), indent=1)
for k, w in list(res['kernels'].items())[:24]:
  print('%-90s %8.1f us x%d' % (k[:90], w['avg_us_under_pmc'], w['launches_seen']), w['derived'])
