#!/bin/bash
# GPU box: SQ / TCC counter passes (separate rocprofv3 --pmc runs) over one probe script; per-launch averages of the kernels whose
# name contains PATTERN.   usage: tools/pmc_kernel.sh PATTERN script.py [args...]      (env is inherited by the script)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PAT=$1; shift
SCRIPT=$1; shift
case $SCRIPT in /*) ;; *) SCRIPT=$R/$SCRIPT;; esac
OUT=/tmp/pmck; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p$i -- python3 "$SCRIPT" "$@" > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/*/*counter_collection.csv'):
  for r in csv.DictReader(open(f)):
    if '$PAT' in r['Kernel_Name']:
      k = r['Kernel_Name'][:70]
      agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for f in glob.glob('$OUT/p1/*/*kernel_trace.csv'):
  for r in csv.DictReader(open(f)):
    if '$PAT' in r['Kernel_Name']:
      dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
  print(k, ' avg %.1f us over %d launches' % (sum(dur[k]) / max(1, len(dur[k])), len(dur[k])))
  for c, v in sorted(d.items()): print('   %-28s %16.0f per launch' % (c, v / cnt[(k, c)]))
PY
