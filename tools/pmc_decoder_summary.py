"""Summarises tools/pmc_decoder.sh's passes: per arithmetic mode, the conv kernel of the decoder-block forward --
HBM bytes per launch ((2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts half of streaming reads,
MI355X_MICROARCH.md) and the SQ counters per launch (MFMA busy, LDS, waits, clock)."""
import collections
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

root, out_traffic, out_sq = sys.argv[1:4]
CONV = ('conv_patch', 'conv_cb8', 'conv16')


def per_kernel(d):
  agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
  dur = collections.defaultdict(lambda: [0, 0.0])
  for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
      k = r['Kernel_Name']
      a = agg[k][r['Counter_Name']]
      a[0] += 1; a[1] += float(r['Counter_Value'])
  for f in glob.glob(d + '/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
      a = dur[r['Kernel_Name']]
      a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
  return agg, dur


traffic, sq = {}, {}
for prec in ('fp32', 'bf16x6', 'bf16'):
  ent, counters = {}, {}
  for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
    agg, dur = per_kernel(os.path.join(root, '%s_%s' % (prec, ctr)))
    convs = {k: v for k, v in agg.items() if any(c in k for c in CONV)}
    if not convs:
      continue
    k = max(convs, key=lambda k: convs[k][ctr][1])
    n, v = convs[k][ctr]
    ent['kernel'] = k[:100]
    ent[ctr.lower() + '_kb_per_launch'] = round(v / n, 1)
    ent['launches'] = n
    # the normalising launch of the block, where BatchNorm is not inside the conv launch (fp32 modes)
    bns = {kk: vv for kk, vv in agg.items() if 'bn_finalize_apply' in kk}
    if bns:
      kb = max(bns, key=lambda kk: bns[kk][ctr][1])
      nb, vb = bns[kb][ctr]
      ent['bn_kernel'] = kb[:100]
      ent['bn_' + ctr.lower() + '_kb_per_launch'] = round(vb / nb, 1)
    if k in dur:
      ent['avg_us_under_pmc'] = round(dur[k][1] / dur[k][0], 2)
  if 'fetch_size_kb_per_launch' in ent and 'write_size_kb_per_launch' in ent:
    ent['hbm_bytes_per_launch'] = int((2 * ent['fetch_size_kb_per_launch'] + ent['write_size_kb_per_launch']) * 1024)
    # the block (conv + BatchNorm + LeakyReLU): the conv launch alone when BatchNorm runs inside it
    ent['block_hbm_bytes'] = ent['hbm_bytes_per_launch'] + int((2 * ent.get('bn_fetch_size_kb_per_launch', 0) +
                                                                ent.get('bn_write_size_kb_per_launch', 0)) * 1024)
    ent['src_hash'] = bench.source_hash()
    traffic[prec] = ent
  for d in sorted(glob.glob(os.path.join(root, prec + '_sq*'))):
    agg, dur = per_kernel(d)
    convs = {k: v for k, v in agg.items() if any(c in k for c in CONV)}
    if not convs:
      continue
    k = max(convs, key=lambda k: sum(a[0] for a in convs[k].values()))
    for c, (n, v) in convs[k].items():
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
      # busy cycles summed over the chip's 1024 SIMDs / (SIMDs x kernel cycles at the nominal 2.4 GHz)
      der['mfma_busy_frac'] = round(w['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * w['avg_us_under_pmc'] * 2400.0), 4)
    if 'SQ_WAVE_CYCLES' in w and w['SQ_WAVE_CYCLES']:
      for k, name in (('SQ_WAIT_ANY', 'wait_frac_of_wave_cycles'), ('SQ_ACTIVE_INST_ANY', 'issue_frac_of_wave_cycles'),
                      ('SQ_WAIT_INST_ANY', 'stall_frac_of_wave_cycles')):
        if k in w:
          der[name] = round(w[k] / w['SQ_WAVE_CYCLES'], 4)
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
