#!/bin/bash
# GPU box: SQ / TCC counter passes (separate rocprofv3 --pmc runs, --kernel-trace only) over eager steps of the bench workload;
# per-launch averages of every kernel -> JSON.   usage: tools/pmc_step.sh PRECISION KIND OUT.json [N]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
PREC=$1; KIND=$2; OUTJ=$3; N=${4:-4}
case $OUTJ in /*) ;; *) OUTJ=$R/$OUTJ;; esac
OUT=/tmp/pmcs_$PREC$KIND; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/p$i -- python3 $R/tools/probe_step_eager.py $PREC $KIND $N > $OUT/log$i.txt 2>&1
  tail -2 $OUT/log$i.txt
done
python3 $R/tools/pmc_step_summary.py $OUT $OUTJ $PREC $KIND
