"""Summarises tools/pmc_chain.sh's passes: per arithmetic mode, the chained decoder launch -- HBM bytes per launch
((2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of streaming reads, MI355X_MICROARCH.md) and the SQ counters per
launch (MFMA busy, LDS, waits, clock)."""
import collections
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

root, out_traffic, out_sq = sys.argv[1:4]
NAMES = {'fp32': 'chain32_kernel', 'bf16': 'chain16_kernel', 'bf16_m4': 'chain16_kernel'}     # key -> kernel; 'bf16_m4' = configs[1] (M = 4)


def per_kernel(d):
  agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
  dur = collections.defaultdict(lambda: [0, 0.0])
  for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
      a = agg[r['Kernel_Name']][r['Counter_Name']]
      a[0] += 1; a[1] += float(r['Counter_Value'])
  for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
      a = dur[r['Kernel_Name']]
      a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
  return agg, dur


traffic, sq = {}, {}
for prec, kname in NAMES.items():
  ent, counters = {}, {}
  for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg, dur = per_kernel(os.path.join(root, '%s_%s' % (prec, ctr)))
    ks = [k for k in agg if kname in k and 'prep' not in k]
    if not ks:
      continue
    k = ks[0]
    n, v = agg[k][ctr]
    ent['kernel'] = k[:100]
    ent[ctr.lower() + '_kb_per_launch'] = round(v / n, 1)
    ent['launches'] = n
    if k in dur:
      ent['avg_us_under_pmc'] = round(dur[k][1] / dur[k][0], 2)
  if 'fetch_size_kb_per_launch' in ent and 'write_size_kb_per_launch' in ent:
    ent['hbm_bytes_per_launch'] = int((2 * ent['fetch_size_kb_per_launch'] + ent['write_size_kb_per_launch']) * 1024)
    ent['what'] = 'decoder.0-3 + logits + softmax mixture in one launch, train mode, y_raw / y / z kept for the backward pass (B=32, M=%d)' % (4 if prec.endswith('_m4') else 8)
    ent['src_hash'] = bench.source_hash()
    traffic[prec] = ent
  for d in sorted(glob.glob(os.path.join(root, prec + '_sq*'))):
    agg, dur = per_kernel(d)
    ks = [k for k in agg if kname in k and 'prep' not in k]
    if not ks:
      continue
    k = ks[0]
    for c, (n, v) in agg[k].items():
      counters[c] = round(v / n, 1)
    if k in dur:
      counters.setdefault('avg_us_under_pmc', round(dur[k][1] / dur[k][0], 2))
    counters['kernel'] = k[:100]
  if counters:
    w = counters
    der = {}
    if 'GRBM_GUI_ACTIVE' in w and w.get('avg_us_under_pmc'):
      der['clock_ghz_est'] = round(w['GRBM_GUI_ACTIVE'] / 8 / (w['avg_us_under_pmc'] * 1e3), 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in w and w.get('avg_us_under_pmc'):
      der['mfma_busy_frac'] = round(w['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * w['avg_us_under_pmc'] * 2400.0), 4)
    if 'SQ_WAVE_CYCLES' in w and w['SQ_WAVE_CYCLES']:
      for k2, name in (('SQ_WAIT_ANY', 'wait_frac_of_wave_cycles'), ('SQ_ACTIVE_INST_ANY', 'issue_frac_of_wave_cycles'),
                       ('SQ_WAIT_INST_ANY', 'stall_frac_of_wave_cycles')):
        if k2 in w:
          der[name] = round(w[k2] / w['SQ_WAVE_CYCLES'], 4)
    if 'SQ_INSTS_VALU' in w and w.get('SQ_WAVES'):
      der['valu_insts_per_wave'] = round(w['SQ_INSTS_VALU'] / w['SQ_WAVES'], 1)
    if 'SQ_LDS_BANK_CONFLICT' in w and w.get('SQ_LDS_IDX_ACTIVE'):
      der['lds_conflict_frac'] = round(w['SQ_LDS_BANK_CONFLICT'] / w['SQ_LDS_IDX_ACTIVE'], 4)
    w['derived'] = der
    w['src_hash'] = bench.source_hash()
    sq[prec] = w
json.dump(traffic, open(out_traffic, 'w'), indent=1)
json.dump(sq, open(out_sq, 'w'), indent=1)
print(json.dumps(traffic, indent=1)); print(json.dumps(sq, indent=1))
