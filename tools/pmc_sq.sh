#!/bin/bash
# GPU box: SQ counter passes over the decoder-shaped conv (tools/probe_one.py g8); prints per-kernel sums
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $R/gpurun_out/pmcsq$i -- python3 $R/tools/probe_one.py ${1:-g8} > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmcsq*/*/*counter_collection.csv'):
  for r in csv.DictReader(open(f)):
    if 'conv_patch' in r['Kernel_Name'] or 'wgrad_patch' in r['Kernel_Name']:
      k = r['Kernel_Name'][:60]
      agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
  print(k)
  for c, v in sorted(d.items()): print('   %-28s %14.0f per launch' % (c, v / cnt[(k, c)]))
PY
