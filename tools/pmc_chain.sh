#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the chained decoder launch alone
# (tools/probe_chain_pmc.py) for fp32 and bf16 -> gpurun_out/pmc_decoder.json, gpurun_out/sq_counters.json
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmcchain
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for prec in ${@:-fp32 bf16}; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$OUT/${prec}_$ctr"
    rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d "$OUT/${prec}_$ctr" -- python3 "$R/tools/probe_chain_pmc.py" $prec > /dev/null 2>&1 || true
  done
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAVES"; do
    i=$((i+1))
    rm -rf "$OUT/${prec}_sq$i"
    rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$OUT/${prec}_sq$i" -- python3 "$R/tools/probe_chain_pmc.py" $prec > /dev/null 2>&1 || true
  done
done
# BASELINE configs[1] (M = S = 4, bf16): traffic passes only
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/bf16_m4_$ctr"
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d "$OUT/bf16_m4_$ctr" -- python3 "$R/tools/probe_chain_pmc.py" bf16 20 4 > /dev/null 2>&1 || true
done
cd "$R" && python3 tools/pmc_chain_summary.py "$OUT" gpurun_out/pmc_decoder.json gpurun_out/sq_counters.json
