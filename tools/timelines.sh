#!/bin/bash
# timelines of one captured step: tools/timelines.sh "fp32 bf16" "G D"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p "$R/gpurun_out"
for P in ${1:-fp32}; do for K in ${2:-G D}; do
rm -rf /tmp/tr$P$K
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$P$K -- python3 "$R/tools/trace_step.py" $P $K 12 > /dev/null 2>&1
python3 "$R/tools/trace_summary.py" "$(ls /tmp/tr$P$K/*/*kernel_trace.csv | head -1)" "$R/gpurun_out/tl_${P}_$K.json" > /dev/null 2>&1
done; done
